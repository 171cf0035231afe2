"""CPU: streams from tests/deflate_writer.py -- a DEFLATE *writer* that emits chosen tokens with chosen code lengths: what no zlib
encoder writes but the reference accepts (13-15-bit codes in long blocks, incomplete codes, a single distance code, code 16 first,
code-length runs past HLIT + HDIST, 258 / 32768 matches, hundreds of tiny blocks at odd bit offsets) -- through the oracle and
through the kernel source's host model, strips included, valid and corrupted.  The GPU side of the same streams is
tests/test_gpu_parity.py::test_exotic_streams_vs_oracle."""
import os
import subprocess
import zlib

import pytest

import corpus
import deflate_writer as W
from conftest import ROOT
from test_model_vs_oracle import _build_model, same

POOR_FLAGS = ["-DPZG_STRIP_BACK=8", "-DPZG_STRIP_ROUNDS=2"]  # the strips' run-up cut to 8 bits, two rounds: most lanes start wrong, spans end early ("poor")


def lab_library(tag, flags):
    """build/lab_<tag>/libpzg.so: the product's sources with extra -D options (tests/tools/lab_build.sh), rebuilt when a source is newer."""
    so = os.path.join(ROOT, "build", "lab_" + tag, "libpzg.so")
    srcs = [os.path.join(ROOT, "pure_zlib_amd", "csrc", f) for f in ("inflate_core.h", "bundle_core.h", "wave.h", "pzg_inflate_kernel.h", "pzg_bundle_kernel.h", "pzg_kernels.hip", "pzg_kernels_b.hip", "pzg_launch.h", "pzg_helpers.h", "pzg_api.cpp",
                                                                    "pzg_errors.cpp", "pzg.map")] + [os.path.join(ROOT, "include", "pzg.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(map(os.path.getmtime, srcs)):
        subprocess.check_call([os.path.join(ROOT, "tests", "tools", "lab_build.sh"), tag, *flags])
    return so


def test_writer_is_a_correct_deflate_writer():
    """The writer itself, against an independent decoder: with Huffman codes and plain run-length headers its streams are
    ordinary DEFLATE, and the system zlib decodes them to the bytes the token generator produced."""
    import random
    for seed in range(12):
        rng = random.Random(seed)
        out = bytearray()
        w = W.BitWriter()
        nblocks = rng.randint(1, 12)
        for i in range(nblocks):
            kind = rng.choice(["dynamic", "dynamic", "fixed", "stored"])
            b = W.Block(kind)
            if kind == "stored":
                b.raw = bytes(rng.getrandbits(8) for _ in range(rng.randint(0, 500)))
                out.extend(b.raw)
            else:
                b.tokens = W.gen_tokens(rng, out, rng.randint(1, 20000), dict(alphabet=W._alphabet(rng, "text"), lens=list(range(3, 259)), dists="any", p_match=0.3))
            W.write_block(w, b, i == nblocks - 1, rng, dict(codes="huffman", rle=rng.choice(["plain", "rle"])))
        data = bytes(out)
        z = bytes([0x78, 0x9c]) + w.bytes() + zlib.adler32(data).to_bytes(4, "big")
        assert zlib.decompress(z) == data, seed


def test_oracle_decodes_exotic_streams(oracle):
    """... and with the exotic choices (which the system zlib rejects, all of them) the oracle -- the restated reference -- decodes
    them to the same bytes, consuming the whole stream."""
    rejected = 0
    for seed in range(96):
        d, z, note = W.exotic_stream(seed)
        r, o = oracle.decompress(z, len(d))
        assert r.status == 0 and o == d and r.in_used == len(z) and r.adler == zlib.adler32(d), (seed, note, r.status, r.message)
        try:
            zlib.decompress(z)
        except zlib.error:
            rejected += 1
    assert rejected >= 90, rejected  # (what the writer is for)


@pytest.fixture(scope="module")
def model():
    return _build_model([])


@pytest.fixture(scope="module")
def model_poor():
    return _build_model(POOR_FLAGS)


def test_model_exotic_streams_vs_oracle(model, oracle):
    """The kernel source's host model (strips, groups of sequences, windows for the tiny blocks) on 24 such streams: rings 11 and
    15 on the valid stream, two corrupted variants on ring 11 and one on ring 13 -- status, detail words, in_used, Adler-32, every byte."""
    for seed in range(24):
        d, z, note = W.exotic_stream(seed)
        ro, oo = oracle.decompress(z, len(d))
        for rb in (11, 15):
            rm, om = model(z, len(d), rb)
            assert same(ro, oo, rm, om), (seed, rb, note, ro.status, rm.status)
        for c, rb in ((0, 11), (1, 11), (2, 13)):
            zc = corpus.corrupt(z, seed * 8 + c)
            cap = [len(d), len(d) + 64, len(d) // 2][c]
            ro2, oo2 = oracle.decompress(zc, cap)
            rm, om = model(zc, cap, rb)
            assert same(ro2, oo2, rm, om), (seed, c, rb, note, ro2.status, rm.status, ro2.message)


def test_model_exotic_streams_when_the_guesses_fail(model_poor, oracle):
    """The same with the strips' run-up cut to 8 bits and two rounds of phase B: spans end after a strip or two, the rest of the
    block goes to the windows -- and an error found inside such a span is the error the oracle finds (the span's status
    is looked at before `poor`)."""
    for seed in range(8):
        d, z, note = W.exotic_stream(seed)
        ro, oo = oracle.decompress(z, len(d))
        rm, om = model_poor(z, len(d), 11)
        assert same(ro, oo, rm, om), (seed, note)
        for c in range(3):
            zc = corpus.corrupt(z, seed * 8 + c)
            ro2, oo2 = oracle.decompress(zc, len(d))
            rm, om = model_poor(zc, len(d), 11)
            assert same(ro2, oo2, rm, om), (seed, c, note, ro2.status, rm.status)


NOTHREADS_FLAGS = ["-DPZG_LAB", "-DPZG_LAB_NO_CALL_THREADS"]  # the host paths' helper threads of a call cannot be started: the stages run inline


def test_lab_library_without_call_threads_builds():
    """build/lab_nothreads/libpzg.so (for tests/test_gpu_api.py::test_host_paths_when_no_thread_can_be_started) builds here and
    travels to the GPU box with the snapshot."""
    so = lab_library("nothreads", NOTHREADS_FLAGS)
    assert b"hipv4-amdgcn-amd-amdhsa--gfx950" in open(so, "rb").read()


PROFSEED_FLAGS = ["-DPZG_LAB", "-DPZG_LAB_SEED_PROFILE"]  # every stream is laid out by a well-marked profile that no stream taught the wave


def test_lab_library_with_seeded_profiles_builds():
    """build/lab_profseed/libpzg.so (for tests/test_gpu_parity.py::test_strips_laid_out_by_fuzzed_profiles_on_the_device) builds here and
    travels to the GPU box with the snapshot."""
    so = lab_library("profseed", PROFSEED_FLAGS)
    assert b"hipv4-amdgcn-amd-amdhsa--gfx950" in open(so, "rb").read()


def test_lab_library_with_failing_guesses_builds():
    """build/lab_poor/libpzg.so (the device build of the same experiment, for tests/test_gpu_parity.py) builds here, carries a
    gfx950 code object, and travels to the GPU box with the snapshot."""
    so = lab_library("poor", POOR_FLAGS)
    assert b"hipv4-amdgcn-amd-amdhsa--gfx950" in open(so, "rb").read()
