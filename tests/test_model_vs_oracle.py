"""CPU: the kernel source itself (pure_zlib_amd/csrc/inflate_core.h) compiled as a host program
(wave.h: one host thread, 64-lane LaneVecs emulated) and checked against the oracle.  This pins
the kernel's control logic, table construction, window/segment logic, hybrid near/far window and
error ordering without a GPU.  The model is test infrastructure; libpzg.so never contains it."""
import ctypes as C
import hashlib
import os
import subprocess
import zlib

import pytest

import corpus
from conftest import REF_CASES, ROOT, read_case
from test_oracle_golden import load_vectors

RINGS = [15, 14, 13, 12, 11]  # every shipped instance (pzg_set_option accepts 11..15)


class R(C.Structure):
    _fields_ = [("status", C.c_int32), ("detail0", C.c_uint32), ("detail1", C.c_uint32), ("adler", C.c_uint32),
                ("out_len", C.c_uint64), ("in_used", C.c_uint64)]


def _build_model(flags):
    d = os.path.join(ROOT, "tests", "model")
    so = os.path.join(d, "libpzgmodel%s.so" % ("_" + hashlib.md5(" ".join(flags).encode()).hexdigest()[:8] if flags else ""))
    srcs = [os.path.join(d, "model_harness.cpp"), os.path.join(ROOT, "pure_zlib_amd", "csrc", "inflate_core.h"),
            os.path.join(ROOT, "pure_zlib_amd", "csrc", "wave.h"), os.path.join(ROOT, "pure_zlib_amd", "csrc", "bundle_core.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(map(os.path.getmtime, srcs)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", *flags, "-o", so, srcs[0]])
    M = C.CDLL(so)
    M.pzm_decompress.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(R)]
    M.pzm_decompress_gzip.argtypes = M.pzm_decompress.argtypes

    def run(z, cap, rb, gzip=False):
        out = C.create_string_buffer(max(cap, 1))
        r = R()
        assert (M.pzm_decompress_gzip if gzip else M.pzm_decompress)(z, len(z), out, cap, rb, C.byref(r)) == 0
        return r, out.raw[: min(r.out_len, cap)]
    return run


@pytest.fixture(scope="session")
def model():
    # PZG_MODEL_FLAGS: extra -D options for the host model
    return _build_model(os.environ.get("PZG_MODEL_FLAGS", "").split())


@pytest.fixture(scope="session")
def model_bad_guesses():
    """The host model with the strips' run-up cut to 8 bits and ONE round of phase B: nearly every lane starts in the wrong place,
    spans end after a strip or two and the rest of the block falls back to the windows (strip_span(): s_poor)."""
    return _build_model(["-DPZG_STRIP_BACK=8", "-DPZG_STRIP_ROUNDS=1"])


def same(ro, oo, rm, om):
    if ro.status != rm.status:
        return False
    if ro.status == 0:
        return oo == om and ro.adler == rm.adler and ro.in_used == rm.in_used and ro.out_len == rm.out_len
    if ro.status == 14:
        return ro.out_len == rm.out_len
    if ro.status in (3, 4, 6, 10, 11, 12, 13):
        return (ro.detail0, ro.detail1) == (rm.detail0, rm.detail1)
    if ro.status == 7:
        return (ro.detail0 & 0xff) == rm.detail0
    return True


@pytest.mark.parametrize("rb", RINGS)
def test_model_reference_gold_files(model, rb):
    for name in REF_CASES:
        z, gold = read_case(name)
        r, out = model(z, len(gold), rb)
        assert r.status == 0 and out == gold and r.in_used == len(z) and r.adler == zlib.adler32(gold), name


@pytest.mark.parametrize("rb", RINGS)
def test_model_pinned_vectors(model, rb):
    for v in load_vectors():
        z = bytes.fromhex(v["z"])
        r, out = model(z, 1 << 21, rb)
        assert r.status == v["status"], v["name"]
        assert r.out_len == v["out_len"] or v["status"] != 0
        if v["status"] == 0:
            assert hashlib.sha256(out).hexdigest() == v["out_sha256"] and r.adler == v["adler"] and r.in_used == v["in_used"]
        if v["status"] in (3, 4, 6, 10, 11, 12, 13):
            assert [r.detail0, r.detail1] == v["detail"], v["name"]


@pytest.mark.parametrize("rb", [15, 12])
def test_model_fuzz_valid_and_corrupt(model, oracle, rb):
    for seed in range(250):
        n = [0, 1, 2, 5, 100, 1000, 5000, 40000, 70000][seed % 9] if seed % 7 == 0 else (seed * 37) % 12000
        d = corpus.mixed_data(n, seed)
        z = corpus.compress_variant(d, seed)
        r, out = model(z, len(d), rb)
        assert r.status == 0 and out == d and r.adler == zlib.adler32(d) and r.in_used == len(z), seed
    for seed in range(1200):
        d = corpus.mixed_data((seed * 131) % 3000 + 1, seed)
        z = corpus.corrupt(corpus.compress_variant(d, seed), seed)
        cap = [len(d), len(d) + 100, 1 << 17][seed % 3]
        ro, oo = oracle.decompress(z, cap)
        rm, om = model(z, cap, rb)
        assert same(ro, oo, rm, om), (seed, ro.status, rm.status, ro.message)


@pytest.mark.parametrize("rb", [14, 13, 12, 11])
def test_model_far_window_and_small_capacity(model, oracle, rb):
    """Back-references older than the LDS ring come from the flushed output; an output larger than
    its capacity is handed to the 32 KiB-ring pass (the harness does what the fixup launch does)."""
    for seed in range(16):
        n = [32768, 65536, 100000, 5000][seed % 4]
        d = corpus.zipf_text(n, seed) if seed % 2 == 0 else (corpus.random_bytes(3000, seed) + corpus.zipf_text(20000, seed) + corpus.random_bytes(3000, seed) * 3)
        z = zlib.compress(d, 1 + seed % 9)
        for cap in (len(d), len(d) // 2, 0):
            ro, oo = oracle.decompress(z, cap)
            rm, om = model(z, cap, rb)
            assert (ro.status, ro.out_len) == (rm.status, rm.out_len) and oo == om, (seed, cap)


@pytest.mark.parametrize("rb", [15, 11])
def test_model_long_codes_second_level_tables(model, oracle, rb):
    """Codes longer than the 8-bit primary table: second-level tables in the windows, the exact walk in
    the checked path; every level and strategy, plus corrupted variants against the oracle."""
    for seed in range(36):
        n = [3000, 20000, 40000][seed % 3]
        d = corpus.skewed_bytes(n, seed) if seed % 2 else corpus.html_slice(n, seed)
        z = corpus.compress_variant(d, seed) if seed % 3 == 0 else zlib.compress(d, 1 + seed % 9)
        r, out = model(z, len(d), rb)
        assert r.status == 0 and out == d and r.adler == zlib.adler32(d) and r.in_used == len(z), seed
        for c in range(6):
            zc = corpus.corrupt(z, seed * 16 + c)
            ro, oo = oracle.decompress(zc, len(d) + 64)
            rm, om = model(zc, len(d) + 64, rb)
            assert same(ro, oo, rm, om), (seed, c, ro.status, rm.status, ro.message)


@pytest.mark.parametrize("rb", [11, 15])
def test_model_binary_records_long_codes_inside_spans(model, oracle, rb):
    """Round 6: 16-byte binary-looking records -- ~250 literals of 8 to 10 bits, more second-level entries than the pool holds (what the
    depth cap leaves of it goes to the first prefixes it cut short), and a token every ~200 that only decode_long() resolves: the
    spans' lanes take those through strip_resolve() instead of stopping.  Valid streams of all three levels and corrupted ones."""
    for seed in range(16):
        d = corpus.binary_records([8, 16, 30, 64][seed % 4] * 1024, seed)
        z = zlib.compress(d, [6, 1, 9][seed % 3])
        r, out = model(z, len(d), rb)
        assert r.status == 0 and out == d and r.adler == zlib.adler32(d) and r.in_used == len(z), seed
        for k in range(6):
            zc = corpus.corrupt(z, 16 * seed + k)
            cap = [len(d), len(d) + 100, len(d) // 2][k % 3]
            ro, oo = oracle.decompress(zc, cap)
            rm, om = model(zc, cap, rb)
            assert same(ro, oo, rm, om), (seed, k, ro.status, rm.status, ro.message)


@pytest.mark.parametrize("rb", [11, 15])
def test_model_codes_around_the_second_level_pool(model, oracle, rb):
    """Writer-made blocks (deflate_writer.pool_stream) whose literal/length codes have 150-256 symbols of 8 to 13 bits in constant
    use: a little less, a little more and much more than the pool's 252 entries -- every depth cap, with and without a remainder for the
    prefixes it cut short (build_table: dup), long codes of literals, lengths and distances met by the spans' lanes (strip_resolve).
    Valid and corrupted, against the oracle."""
    import deflate_writer as W
    for seed in range(96):
        d, z = W.pool_stream(seed)
        ro, oo = oracle.decompress(z, len(d))
        rm, om = model(z, len(d), rb)
        assert ro.status == 0 and oo == d and same(ro, oo, rm, om), (seed, ro.status, rm.status)
        for k in range(4):
            zc = corpus.corrupt(z, 8 * seed + k)
            cap = [len(d), len(d) + 100, len(d) // 2][k % 3]
            ro, oo = oracle.decompress(zc, cap)
            rm, om = model(zc, cap, rb)
            assert same(ro, oo, rm, om), (seed, k, ro.status, rm.status, ro.message)


@pytest.mark.parametrize("rb,strips", [(11, True), (15, True), (12, True), (11, False)])
def test_model_strips(model, oracle, rb, strips, monkeypatch):
    """Round 4: long runs of input are decoded by strips (64 lanes, one piece of the input each, from speculative starts that are
    verified and repaired; inflate_core.h strip_span()).  Streams long enough for several spans -- one block and many, every
    strategy, two-bit codes -- and corrupted variants of each against the oracle: status, detail, in_used, every byte.  The
    same streams once more with the scratch withheld (a launch whose scratch could not be allocated): the windows alone."""
    if not strips:
        monkeypatch.setenv("PZM_NO_STRIPS", "1")
    for seed in range(30 if strips else 12):
        d, z = corpus.strip_case(seed)
        r, out = model(z, len(d), rb)
        assert r.status == 0 and out == d and r.adler == zlib.adler32(d) and r.in_used == len(z), seed
        for c in range(6):
            zc = corpus.corrupt(z, seed * 16 + c)
            cap = [len(d) + 64, len(d) // 2][c % 2] if c < 4 else len(d)
            ro, oo = oracle.decompress(zc, cap)
            rm, om = model(zc, cap, rb)
            assert same(ro, oo, rm, om), (seed, c, ro.status, rm.status, ro.message)


@pytest.mark.parametrize("rb", [11, 15])
def test_model_strips_when_the_guesses_fail(model_bad_guesses, oracle, rb):
    """Nothing about the result may depend on the strips' speculative starts: with the run-up and the repair rounds all but
    switched off the same streams and corruptions decode to the same results (and the block goes on by windows)."""
    for seed in range(18):
        d, z = corpus.strip_case(seed)
        r, out = model_bad_guesses(z, len(d), rb)
        assert r.status == 0 and out == d and r.adler == zlib.adler32(d) and r.in_used == len(z), seed
        for c in range(4):
            zc = corpus.corrupt(z, seed * 16 + c)
            ro, oo = oracle.decompress(zc, len(d) + 64)
            rm, om = model_bad_guesses(zc, len(d) + 64, rb)
            assert same(ro, oo, rm, om), (seed, c, ro.status, rm.status, ro.message)


@pytest.fixture(scope="session")
def model_few_steps():
    """The host model with 80 steps (and tokens) per lane and span: strips run out of steps, spans end at the lane that did
    (strip_span(): STF 2, `tight`), and the wave's profile is forgotten and learnt again all the time."""
    return _build_model(["-DPZG_STRIP_TMAX=80", "-DPZG_PROF_CMIN=256"])


@pytest.mark.parametrize("which,rb", [("model", 11), ("model", 15), ("few", 11), ("few", 13)])
def test_model_strips_laid_out_by_the_last_streams_profile(model, model_few_steps, oracle, which, rb):
    """Round 5: a wave remembers where the tokens of the last stream's first span lay (strip_profile_learn: in its scratch, which
    the model keeps from call to call as a persistent wave does) and cuts the next stream's strips there, unless the run-ups
    disagree (strip_profile_check).  Nothing about the result may depend on it: 90 streams in a row whose neighbours differ in kind,
    size, level and validity -- every one against the oracle; and the same with lanes that run out of steps all the time."""
    import numpy as np
    run = model if which == "model" else model_few_steps
    rng = np.random.default_rng(5)
    kinds = []
    for i in range(16):
        size = int(rng.choice([12288, 24576, 32768, 49152]))
        d = [corpus.zipf_text, corpus.html_slice, corpus.skewed_bytes, corpus.mixed_data][i % 4](size, i)
        kinds.append((d, zlib.compress(d, [6, 1, 9, 6, 3][(i // 4) % 5])))
    seq = list(rng.integers(0, len(kinds), 60)) + [0] * 10 + [1, 2] * 10  # (a run of one kind: the profile settles; then two kinds in turn)
    for n, j in enumerate(seq[: 90 if which == "model" else 50]):
        d, z = kinds[j]
        if n % 7 == 3:
            zc = corpus.corrupt(z, n)
            ro, oo = oracle.decompress(zc, len(d))
            rm, om = run(zc, len(d), rb)
            assert same(ro, oo, rm, om), (n, j, ro.status, rm.status, ro.message)
        else:
            r, out = run(z, len(d), rb)
            assert r.status == 0 and out == d and r.adler == zlib.adler32(d) and r.in_used == len(z), (n, j)


@pytest.mark.parametrize("rb", [15, 14, 11])
def test_model_gzip_members(model, oracle, rb):
    """The gzip extension (RFC 1952 header / CRC-32 + ISIZE trailer around the same DEFLATE core): the kernel
    source against the oracle's gzip restatement, valid and corrupted."""
    for seed in range(60):
        d = corpus.mixed_data((seed * 613) % 30000, seed)
        z = corpus.gzip_member(d, seed)
        r, out = model(z, len(d) + 8, rb, gzip=True)
        assert r.status == 0 and out == d and r.adler == zlib.crc32(d) and r.in_used == len(z), (seed, r.status)
        for c in range(8):
            zc = corpus.corrupt(z, seed * 64 + c)
            ro, oo = oracle.gzip_decompress(zc, len(d) + 8)
            rm, om = model(zc, len(d) + 8, rb, gzip=True)
            if rm.status == 14 and ro.status in (10, 19):
                # documented: an output larger than its capacity is not stored, so its CRC-32 / ISIZE are not
                # verified (status 14 carries the size to retry with); the oracle tracks the CRC regardless
                assert rm.out_len == ro.out_len
                continue
            assert ro.status == rm.status, (seed, c, ro.status, rm.status, ro.message)
            if ro.status == 0:
                assert oo == om and ro.adler == rm.adler
            elif ro.status in (10, 18, 19):
                assert (ro.detail0, ro.detail1) == (rm.detail0, rm.detail1) or ro.status == 18


# ---- the resumable instance (decompressIncremental) and the two format extensions, still on the CPU ---------------

class ModelDecoder:
    """Drives pzm_resume_feed the way the host mirror drives pzg_decoder_feed: the unconsumed tail goes in front of the next
    piece, a call that ran out of room is repeated, 32 KiB chunks are published as the device-side count says."""

    def __init__(self, M, room, rb=15):
        self.M, self.room, self.rb = M, room, rb
        self.state = C.create_string_buffer(M.pzm_resume_state_bytes_rb(rb))
        self.tail, self.pending, self.published, self.total = b"", bytearray(), 0, bytearray()
        self.events = [("NeedMore",)]

    def feed(self, piece):
        if not piece:
            self.events.append(("NeedMore",))
            return True
        data = self.tail + piece
        while True:
            out = C.create_string_buffer(self.room)
            r, ch = R(), C.c_uint32(0)
            assert self.M.pzm_resume_feed_rb(self.rb, self.state, data, len(data), 0, out, self.room, C.byref(r), C.byref(ch)) == 0
            self.pending += out.raw[:r.out_len]
            self.total += out.raw[:r.out_len]
            while self.published < ch.value:
                assert len(self.pending) >= 32768
                self.events.append(("Chunk", 32768))
                del self.pending[:32768]
                self.published += 1
            data = data[r.in_used:]
            if r.status != 102:  # (102: out of room -- same input again)
                break
        self.tail = data
        if r.status == 101:
            self.events.append(("NeedMore",))
            return True
        if r.status == 0:
            self.events += [("Chunk", len(self.pending)), ("Done",)]
        else:
            self.events.append(("DecompError", r.status))
        return False


@pytest.fixture(scope="session")
def model_lib(model):
    flags = os.environ.get("PZG_MODEL_FLAGS", "").split()
    M = C.CDLL(os.path.join(ROOT, "tests", "model", "libpzgmodel%s.so" % ("_" + hashlib.md5(" ".join(flags).encode()).hexdigest()[:8] if flags else "")))
    M.pzm_resume_state_bytes_rb.restype = C.c_uint32
    M.pzm_resume_state_bytes_rb.argtypes = [C.c_int]
    M.pzm_resume_feed_rb.argtypes = [C.c_int, C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(R), C.POINTER(C.c_uint32)]
    M.pzm_decompress_dict.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(R)]
    return M


# (the resumable kernel is built for ring 12 -- PZG_RES_RING: a small LDS ring + the decoder's 32 KiB history in HBM; the
# 32 KiB LDS ring and ring 11 are the same template and stay tested)
@pytest.mark.parametrize("rb", [12, 15, 11])
def test_model_incremental_event_trace(model_lib, oracle, rb):
    """The ZlibDecoder protocol (Monad.hs:163-197, OutputWindow.hs:45-54): NeedMore / Chunk(32768) / Chunk(rest) / Done /
    DecompError in exactly the reference's order, for the nine fixtures and seeded streams, fed 7000 bytes, 1 byte, 100
    bytes at a time, with empty pieces, corrupted streams and output rooms from 4 KiB up."""
    cases = []
    for name in REF_CASES:
        z, _gold = read_case(name)
        cases += [(z, 7000, 70000), (z[:3000], 1, 4096)]
    for seed in range(120):
        n = [0, 1, 100, 5000, 70000, 200000][seed % 6]
        d = corpus.mixed_data(n, seed) if seed % 3 else corpus.zipf_text(n, seed)
        z = corpus.compress_variant(d, seed) if seed % 2 else zlib.compress(d, 1 + seed % 9)
        if seed % 11 == 0:
            z = corpus.corrupt(z, seed)
        step = [1, 7, 100, 7000, 50000][seed % 5] if len(z) < 2000 or seed % 5 else 7000
        cases.append((z, step, [4096, 70000, 300000][seed % 3]))
    for k, (z, step, room) in enumerate(cases):
        pieces = [z[i:i + step] for i in range(0, len(z), step)]
        if k % 13 == 0:
            pieces.insert(len(pieces) // 2, b"")
        eo, ro, oo = oracle.trace(pieces)
        dec = ModelDecoder(model_lib, room, rb)
        for p in pieces:
            if not dec.feed(p):
                break
        assert dec.events == eo, (k, len(z), step, room, ro.status)
        if ro.status == 0:
            assert bytes(dec.total) == oo


@pytest.mark.parametrize("rb", [12, 15])
def test_model_incremental_spans_of_strips(model_lib, oracle, rb):
    """Round 5: the resumable decoder takes the strips too -- a span is decoded and emitted inside one call, cut behind the last lane
    whose output still fits the call's room, its far matches read the decoder's history, the reference's chunk count is kept group by
    group.  The bench's shape (256 KiB of text in 32 KiB pieces, 192 KiB rooms), rooms that cut spans short (24 KiB, 9 KiB), a piece
    size that leaves tails for the windows, writer-made streams (tests/deflate_writer.py) and corrupted ones: whole event traces and
    every byte against the oracle's."""
    import deflate_writer as W
    cases = []
    for seed in range(3):
        z = zlib.compress(corpus.zipf_text(256 * 1024, 7000 + seed), 6)
        cases += [(z, 32768, 192 * 1024), (z, 32768, 24 * 1024), (z, 20000, 9 * 1024)]
    for seed in (0, 3, 5, 7):
        d, z, _ = W.exotic_stream(seed)
        cases += [(z, 32768, 70000), (corpus.corrupt(z, seed), 16000, 40000)]
    z = zlib.compress(corpus.html_slice(60000, 1) + corpus.skewed_bytes(90000, 2), 6)
    cases += [(z, 50000, 100000), (corpus.corrupt(z, 5), 50000, 100000)]
    for k, (z, step, room) in enumerate(cases):
        pieces = [z[i:i + step] for i in range(0, len(z), step)]
        eo, ro, oo = oracle.trace(pieces)
        dec = ModelDecoder(model_lib, room, rb)
        for p in pieces:
            if not dec.feed(p):
                break
        assert dec.events == eo, (k, len(z), step, room, ro.status)
        if ro.status == 0:
            assert bytes(dec.total) == oo


@pytest.mark.parametrize("rb", [12, 15])
def test_model_incremental_binary_records(model_lib, oracle, rb):
    """... and the resumable instance's spans (pieces large enough for them, rooms that cut them): whole event traces."""
    for seed in range(6):
        d = corpus.binary_records([16, 30, 64][seed % 3] * 1024, seed)
        z = zlib.compress(d, [6, 1, 9][seed % 3])
        for piece, room in ((8192, 1 << 20), (20000, 9 * 1024), (len(z), 40000)):
            pieces = [z[i:i + piece] for i in range(0, len(z), piece)]
            eo, ro, oo = oracle.trace(pieces)
            dec = ModelDecoder(model_lib, room, rb)
            for pc in pieces:
                if not dec.feed(pc):
                    break
            assert ro.status == 0 and dec.events == eo, (seed, piece, room)
            assert bytes(dec.total) == oo == d


@pytest.mark.parametrize("rb", [12, 15])
def test_model_incremental_resumes_mid_block_with_long_distance_codes(model_lib, oracle, rb):
    """a call that resumes in the middle of a block runs strip_span(), which asks whether the DISTANCE code has second-level
    tables (dist_sub_used) -- a word the ResumeState did not carry.  Writer-made streams whose distance codes are 13-15 bits long in
    constant use, in pieces of 5,000 and 9,000 bytes: every call but the first resumes inside the one long block, with input enough
    for a span; whole event traces and every byte against the oracle's."""
    import deflate_writer as W
    for seed in (0, 8, 16, 1, 9):
        d, z, _ = W.exotic_stream(seed)
        for step, room in ((5000, 70000), (9000, 40000)):
            pieces = [z[i:i + step] for i in range(0, len(z), step)]
            eo, ro, oo = oracle.trace(pieces)
            dec = ModelDecoder(model_lib, room, rb)
            for p in pieces:
                if not dec.feed(p):
                    break
            assert dec.events == eo, (seed, step, room, ro.status)
            assert ro.status == 0 and bytes(dec.total) == oo == d


@pytest.mark.parametrize("rb", [12, 15])
def test_model_incremental_several_chunks_inside_one_group(model_lib, oracle, rb):
    """the reference runs moveWindow after every MATCH (Deflate.hs:106-120) and hands out ONE 32 KiB chunk per call
    (OutputWindow.hs:45-54) -- literals only add to its window.  A block of ~58 KiB with matches, then 44 KiB of nothing but literals,
    then a few matches close together -- the window holds ~100 KiB when they come, and the first TWO of them hand out a chunk each,
    inside one group of the strips' emit -- and nothing but literals from there to the end of the stream: no later match makes up for a
    chunk the group did not count.  Whole event traces against the oracle's."""
    import random
    import deflate_writer as W
    for seed in range(6):
        rng = random.Random(0xC4 + seed)
        out = bytearray()
        b = W.Block("dynamic")
        alpha = W._alphabet(rng, "text")
        b.tokens = W.gen_tokens(rng, out, 56000 + 1500 * seed, dict(alphabet=alpha, lens=[3, 4, 5, 8, 20], dists="near", p_match=0.4))
        b.tokens += W.gen_tokens(rng, out, 44000 + 1000 * seed, dict(alphabet=alpha, lens=[3], dists="near", p_match=0.0))
        b.tokens += W.gen_tokens(rng, out, 12 + 5 * seed, dict(alphabet=alpha, lens=[3, 4, 6], dists="near", p_match=1.0))
        b.tokens += W.gen_tokens(rng, out, 6000 + 4000 * (seed & 1), dict(alphabet=alpha, lens=[3], dists="near", p_match=0.0))
        b.opts = {}
        w = W.BitWriter()
        W.write_block(w, b, True, rng, dict(codes="huffman", rle="rle"))
        d = bytes(out)
        z = bytes([0x78, 0x9c]) + w.bytes() + zlib.adler32(d).to_bytes(4, "big")
        # (the end of a block looks at the window as well, Deflate.hs:47: the first piece ends inside the literals behind the matches)
        for k in (600, 1500, 2800):
            pieces = [z[:len(z) - k], z[len(z) - k:]]
            eo, ro, oo = oracle.trace(pieces)
            assert [e[0] for e in eo] == ["NeedMore", "Chunk", "Chunk", "NeedMore", "Chunk", "Done"]
            dec = ModelDecoder(model_lib, 300000, rb)
            for pc in pieces:
                if not dec.feed(pc):
                    break
            assert ro.status == 0 and dec.events == eo, (seed, len(pieces), dec.events, eo)
            assert bytes(dec.total) == oo == d


def test_model_incremental_bad_header_with_fdict_bit(model_lib, oracle):
    """Zlib.hs:53-67: CMF and FLG are read and checked (FCHECK, method, window) before anything else; a bad header whose FDICT
    bit is set is a DecompError after two bytes, not a NeedMore waiting for a DICTID."""
    def hdr(cmf, flg_hi):  # FLG with FDICT set and a valid FCHECK
        flg = (flg_hi & 0xc0) | 0x20
        flg += (31 - ((cmf << 8) | flg) % 31) % 31
        return bytes([cmf, flg])
    good = hdr(0x78, 0x80)
    assert (good[0] * 256 + good[1]) % 31 == 0 and good[1] & 0x20
    cases = [bytes([0x78, 0x21]),          # FCHECK fails (0x7821 % 31 != 0), FDICT bit set
             hdr(0x77, 0x80),              # method 7
             hdr(0x88, 0x80),              # window 8
             good,                         # a sound FDICT header: NeedMore until the DICTID and more have come
             good + b"\x01\x02\x03"]
    for z in cases:
        for step in (1, 2, 5):
            pieces = [z[i:i + step] for i in range(0, len(z), step)]
            eo, _ro, _oo = oracle.trace(pieces)
            dec = ModelDecoder(model_lib, 4096, 12)
            for p in pieces:
                if not dec.feed(p):
                    break
            assert dec.events == eo, (z.hex(), step, dec.events, eo)


def test_model_preset_dictionary(model_lib, oracle):
    """PZG_FDICT extension: dictionary installed as history (ring 15 instance) against zlib and the oracle."""
    for seed in range(40):
        zd = corpus.zipf_text([5, 300, 20000, 32768, 40000][seed % 5], seed + 100)
        d = (zd[-200:] if seed % 2 else b"") + corpus.zipf_text(1000 + 997 * seed, seed)
        co = zlib.compressobj(6, zdict=zd)
        z = co.compress(d) + co.flush()
        out = C.create_string_buffer(len(d) + 64)
        r = R()
        model_lib.pzm_decompress_dict(z, len(z), zd, len(zd), out, len(d) + 64, C.byref(r))
        ro, oo = oracle.decompress_dict(z, zd)
        assert r.status == 0 and out.raw[:r.out_len] == d == oo and ro.status == 0 and r.adler == zlib.adler32(d), seed
        do = zlib.decompressobj(zdict=zd)
        assert do.decompress(z) == d
        bad = zd[:-1] + bytes([zd[-1] ^ 1])
        model_lib.pzm_decompress_dict(z, len(z), bad, len(bad), out, len(d) + 64, C.byref(r))
        ro, _ = oracle.decompress_dict(z, bad)
        assert r.status == ro.status == 20 and (r.detail0, r.detail1) == (ro.detail0, ro.detail1)


def test_model_gzip_multi_member(model, oracle):
    import gzip
    for seed in range(30):
        parts = [corpus.mixed_data((seed * 977 + 131 * k) % 40000, seed + k) for k in range(1 + seed % 4)]
        g = b"".join(corpus.gzip_member(p, seed + k) if k % 2 else gzip.compress(p, 1 + seed % 9) for k, p in enumerate(parts))
        d = b"".join(parts)
        assert gzip.decompress(g) == d
        for rb in (15, 11):
            r, out = model(g, len(d) + 8, rb, gzip=True)
            assert r.status == 0 and out == d and r.adler == zlib.crc32(d) and r.in_used == len(g), (seed, rb, r.status)
        for c in range(6):
            gc = corpus.corrupt(g, seed * 32 + c)
            ro, oo = oracle.gzip_decompress(gc, len(d) + 8)
            rm, om = model(gc, len(d) + 8, 15, gzip=True)
            if rm.status == 14 and ro.status in (10, 19):
                continue
            assert ro.status == rm.status, (seed, c, ro.status, rm.status, ro.message)
            if ro.status in (10, 19):
                assert (ro.detail0, ro.detail1) == (rm.detail0, rm.detail1)


@pytest.mark.parametrize("rb", [15, 11])
def test_model_queue_never_overfills(model, rb):
    """Small level-6 blobs: a window that ends at a stopper may leave the token queue one short of full, and the token
    after the stopper is then pushed by the checked path -- the queue must still hold at most QCAP tokens (round 2: the
    segment's stop mask is built from `(1 << qn) - 1`; blob 738 of the 2 KiB pool decoded wrong in 1 launch of 5).
    The host model traps in queue_push() if the invariant breaks."""
    for seed in list(range(700, 780)) + list(range(0, 4096, 37)):
        d = corpus.zipf_text(2048, seed)
        z = zlib.compress(d, 6)
        r, out = model(z, len(d), rb)
        assert r.status == 0 and out == d and r.adler == zlib.adler32(d) and r.in_used == len(z), seed


def test_model_profile_fuzz(model, oracle):
    """the wave's profile (strip_profile_layout / _check / _learn) lives in scratch that outlives streams and
    launches and is trusted when its magic word is there.  Before every stream the profile words are overwritten -- random words; the
    magic with random, non-monotone, constant, decreasing or extreme quantiles, extents and token counts; a sound profile stretched or
    squeezed -- and the stream must still decode to the oracle's bytes: a layout can cost rounds, never a token."""
    import random
    M = C.CDLL(os.path.join(ROOT, "tests", "model", "libpzgmodel.so"))
    M.pzm_profile_magic.restype = C.c_uint32
    M.pzm_poke_profile.argtypes = [C.c_int, C.POINTER(C.c_uint32)]
    magic = M.pzm_profile_magic()
    rng = random.Random(0xF022)
    pool = []
    for seed in range(24):
        n = [20000, 32768, 50000, 70000][seed % 4]
        d = [corpus.zipf_text, corpus.html_slice, corpus.skewed_bytes][seed % 3](n, seed)
        pool.append((d, zlib.compress(d, [6, 1, 9][seed % 3])))
    exp = [oracle.decompress(z, len(d)) for d, z in pool]
    for it in range(1000):
        kind = it % 8
        q = [0] * 80
        xt = rng.choice([64, 500, 8000, 30000, 100000, 1 << 18, (1 << 18) + 1, 0xffffffff])
        if kind == 0:
            q = [rng.getrandbits(32) for _ in range(80)]
        elif kind == 1:  # random quantiles under the magic
            q[:64] = [rng.randrange(0, 1 << 18) for _ in range(64)]
        elif kind == 2:  # sorted, but starting anywhere / with plateaus
            q[:64] = sorted(rng.randrange(0, xt % (1 << 19) + 1) for _ in range(64))
            if rng.random() < 0.5:
                q[0] = 0
        elif kind == 3:  # decreasing
            q[:64] = sorted((rng.randrange(0, 1 << 17) for _ in range(64)), reverse=True)
        elif kind == 4:  # constant
            q[:64] = [rng.choice([0, 1, 4095, 1 << 17])] * 64
        elif kind == 5:  # a sound shape, squeezed into the first bits or stretched far beyond any stream
            scale = rng.choice([1, 3, 100, 4000])
            q[:64] = [k * scale for k in range(64)]
        elif kind == 6:  # one quantile out of order in an otherwise sound profile
            q[:64] = [k * 1000 for k in range(64)]
            j = rng.randrange(1, 64)
            q[j] = rng.choice([0, q[j - 1] - 1, 0xffffffff, 1 << 31])
        else:            # zeros
            pass
        if kind != 0:
            q[64] = rng.choice([xt, q[63] + 1, q[63], q[63] + rng.randrange(1, 5000)]) & 0xffffffff
            q[65] = rng.choice([0, 63, 64, 5000, 64 * 192, 64 * 192 + 1, 0xffffffff])
            q[66] = magic
            q[67] = rng.choice([0, 0, 0, 1, 0xffffffff])
            q[68] = rng.choice([0, 0, 31, 32, 0xffffffff])
        arr = (C.c_uint32 * 80)(*[w & 0xffffffff for w in q])
        rb = 11 if it % 5 else 15
        assert M.pzm_poke_profile(rb, arr) == 0
        k = rng.randrange(len(pool))
        d, z = pool[k]
        r, out = model(z, len(d), rb)
        ro, oo = exp[k]
        assert same(ro, oo, r, out) and out == d, (it, kind, k, rb, r.status)
