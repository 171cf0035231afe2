"""GPU: the resumable decoder (decompressIncremental / ZlibDecoder, Monad.hs:163-197) and the two format extensions
(preset dictionaries, multi-member gzip) through the C ABI.

The incremental tests assert EVENT-TRACE equality with the oracle: the exact sequence of NeedMore / Chunk(length) /
Done / DecompError the reference's decoder goes through when it is fed the same pieces (oracle.trace restates
Monad.hs:185-197, 338-358 and OutputWindow.hs:45-60), for the nine fixtures with 7,000-byte and 1-byte feeds, seeded
streams, late errors and empty pieces -- one launch per feed, many decoders per launch."""
import gzip
import zlib

import numpy as np
import pytest

import corpus
from conftest import REF_CASES, read_case

pytestmark = pytest.mark.gpu


def drive(pool, k, pieces):
    """Feed decoder k piece by piece; the events it goes through and the bytes of its chunks."""
    from pure_zlib_amd.incremental import Chunk, DecompError, Done, NeedMore
    events, data = [("NeedMore",)], bytearray()
    st = pool.start(k)
    for p in pieces:
        st = st.feed(p)
        while isinstance(st, Chunk):
            events.append(("Chunk", len(st.chunk)))
            data += st.chunk
            st = st.next()
        if isinstance(st, NeedMore):
            events.append(("NeedMore",))
        elif isinstance(st, Done):
            events.append(("Done",))
            break
        else:
            assert isinstance(st, DecompError)
            events.append(("DecompError", st.error.status))
            return events, bytes(data), st.error
    return events, bytes(data), None


def test_event_trace_reference_fixtures(gpu_ctx, oracle):
    from pure_zlib_amd.incremental import DecoderPool
    for name in REF_CASES:
        z, gold = read_case(name)
        for step in (7000, 1 if len(z) < 2500 else 997):
            pieces = [z[i:i + step] for i in range(0, len(z), step)]
            eo, ro, oo = oracle.trace(pieces)
            pool = DecoderPool(1, gpu_ctx)
            em, data, err = drive(pool, 0, pieces)
            pool.close()
            assert em == eo, (name, step, em[-4:], eo[-4:])
            assert data == gold == oo and err is None


def test_event_trace_seeded_streams_late_errors_small_rooms(gpu_ctx, oracle):
    import pure_zlib_amd.zlib as Z
    from pure_zlib_amd.incremental import DecoderPool
    for seed in range(90):
        n = [0, 1, 100, 5000, 70000, 200000, 400000][seed % 7]
        d = corpus.mixed_data(n, seed) if seed % 3 else corpus.zipf_text(n, seed)
        z = corpus.compress_variant(d, seed) if seed % 2 else zlib.compress(d, 1 + seed % 9)
        if seed % 5 == 0 and len(z) > 40:  # a late error: everything before it must still come out, in order
            z = z[:len(z) - 20] + bytes([z[-20] ^ 0x55]) + z[len(z) - 19:]
        step = [1, 7, 100, 7000, 50000][seed % 5] if len(z) < 1500 or seed % 5 else 7000
        pieces = [z[i:i + step] for i in range(0, len(z), step)]
        if seed % 13 == 0:
            pieces.insert(len(pieces) // 2, b"")
        eo, ro, oo = oracle.trace(pieces)
        pool = DecoderPool(1, gpu_ctx, room=[4096, 70000, 262144][seed % 3])
        em, data, err = drive(pool, 0, pieces)
        pool.close()
        assert em == eo, (seed, n, len(z), step, ro.status, em[-3:], eo[-3:])
        if ro.status == 0:
            assert data == oo == d
        elif err is not None:
            assert err.show() == ro.message.decode(), (seed, err.show(), ro.message)
            assert isinstance(err, Z.DecompressionError)


def test_event_trace_spans_of_strips_and_the_scratch_cap(gpu_ctx, oracle):
    """Round 5: the resumable kernel decodes by strips too (a span inside one call, cut to the call's room, far matches from the
    decoder's history, the reference's chunk count kept per group).  The bench's shape (256 KiB of text in 32 KiB pieces, 192 KiB
    rooms), rooms that cut spans short, writer-made streams (tests/deflate_writer.py), html + literal-heavy data and corrupted
    variants, binary-looking records -- 30 decoders fed in lockstep, one launch per feed -- with the library's scratch unlimited, capped at 32 MiB
    (PZG_OPT_SCRATCH_BYTES: some stream-waves get no slice) and withheld altogether (a 1-byte cap: the windows alone): the same
    event traces and bytes as the oracle every time."""
    import deflate_writer as W
    from pure_zlib_amd.incremental import Chunk, DecoderPool, DecompError, Done, NeedMore
    zs = []
    for seed in range(8):
        zs.append(zlib.compress(corpus.zipf_text(256 * 1024, 7000 + seed), 6))
    for seed in (0, 1, 3, 5, 6, 7):
        z = W.exotic_stream(seed)[1]
        zs += [z, corpus.corrupt(z, seed)]
    z = zlib.compress(corpus.html_slice(60000, 1) + corpus.skewed_bytes(90000, 2), 6)
    zs += [z, corpus.corrupt(z, 5), zlib.compress(corpus.mixed_data(300000, 4), 9), zlib.compress(corpus.zipf_text(100000, 3), 1)]
    # (round 6: binary-looking records -- long codes in constant use, resolved inside the spans: strip_resolve())
    for seed in range(3):
        z = zlib.compress(corpus.binary_records(160 * 1024, seed), [6, 1, 9][seed])
        zs += [z, corpus.corrupt(z, 40 + seed)]
    n = len(zs)
    try:
        for cap, step, room in ((0, 32768, 192 * 1024), (32 << 20, 32768, 24 * 1024), (1, 20000, 70000), (0, 50000, 9 * 1024)):
            gpu_ctx.set_scratch_bytes(cap)
            pool = DecoderPool(n, gpu_ctx, room=room)
            events = [[("NeedMore",)] for _ in range(n)]
            got = [bytearray() for _ in range(n)]
            live, pos = list(range(n)), 0
            while live:
                sts = pool.feed(live, [zs[k][pos:pos + step] for k in live])
                nxt = []
                for k, st in zip(live, sts):
                    while isinstance(st, Chunk):
                        events[k].append(("Chunk", len(st.chunk)))
                        got[k] += st.chunk
                        st = st.next()
                    if isinstance(st, NeedMore):
                        events[k].append(("NeedMore",))
                        if pos + step < len(zs[k]):
                            nxt.append(k)
                    elif isinstance(st, Done):
                        events[k].append(("Done",))
                    else:
                        assert isinstance(st, DecompError)
                        events[k].append(("DecompError", st.error.status))
                live = nxt
                pos += step
            pool.close()
            for k in range(n):
                pieces = [zs[k][i:i + step] for i in range(0, len(zs[k]), step)]
                eo, ro, oo = oracle.trace(pieces)
                assert events[k] == eo, (cap, step, room, k, ro.status, events[k][-3:], eo[-3:])
                if ro.status == 0:
                    assert bytes(got[k]) == oo, (cap, step, room, k)
    finally:
        gpu_ctx.set_scratch_bytes(0)


def test_many_decoders_per_launch(gpu_ctx, oracle):
    """64 decoders fed in lockstep, each launch continuing all that still want input: the traces stay the reference's."""
    from pure_zlib_amd.incremental import Chunk, DecoderPool, Done, NeedMore
    n = 64
    datas = [corpus.zipf_text(20000 + 3001 * k, k) for k in range(n)]
    zs = [zlib.compress(d, 1 + k % 9) for k, d in enumerate(datas)]
    step = 5000
    pool = DecoderPool(n, gpu_ctx)
    events = [[("NeedMore",)] for _ in range(n)]
    got = [bytearray() for _ in range(n)]
    live = list(range(n))
    pos = 0
    while live:
        sts = pool.feed(live, [zs[k][pos:pos + step] for k in live])
        nxt = []
        for k, st in zip(live, sts):
            while isinstance(st, Chunk):
                events[k].append(("Chunk", len(st.chunk)))
                got[k] += st.chunk
                st = st.next()
            if isinstance(st, NeedMore):
                events[k].append(("NeedMore",))
                nxt.append(k)
            else:
                assert isinstance(st, Done)
                events[k].append(("Done",))
        live = nxt
        pos += step
    pool.close()
    for k in range(n):
        pieces = [zs[k][i:i + step] for i in range(0, len(zs[k]), step)]
        eo, ro, oo = oracle.trace(pieces)
        assert events[k] == eo and bytes(got[k]) == datas[k], k


def test_preset_dictionaries_extension(gpu_ctx, oracle):
    """PZG_FDICT-style dictionaries (pzg_decompress_many_dict): against zlib.decompressobj(zdict=...) and the oracle; a
    stream without FDICT ignores its dictionary, the plain call keeps the reference's behaviour (DICTID skipped)."""
    import pure_zlib_amd as P
    streams, dicts, datas = [], [], []
    for seed in range(60):
        zd = corpus.zipf_text([5, 300, 20000, 32768, 40000][seed % 5], seed + 100)
        d = (zd[-200:] if seed % 2 else b"") + corpus.zipf_text(1000 + 997 * seed, seed)
        if seed % 7 == 3:  # no FDICT in the stream: the dictionary supplied for it is not used
            z = zlib.compress(d, 6)
        else:
            co = zlib.compressobj(6, zdict=zd)
            z = co.compress(d) + co.flush()
        streams.append(z)
        dicts.append(zd if seed % 11 else None)
        datas.append(d)
    rs = P.decompress_many(streams, ctx=gpu_ctx, zdict=dicts)
    for k, r in enumerate(rs):
        if dicts[k] is None and (streams[k][1] & 0x20):  # FDICT set but nothing supplied: the reference's path, empty history
            ro, oo = oracle.decompress(streams[k], len(datas[k]) + 64)
            assert (r == P.Right(oo)) if ro.status == 0 else (r.value.show() == ro.message.decode()), k
        else:
            assert r == P.Right(datas[k]), k
    wrong = [bytes([b[0] ^ 1]) + b[1:] if b else b for b in dicts]
    rs = P.decompress_many(streams, ctx=gpu_ctx, zdict=wrong)
    for k, r in enumerate(rs):
        if wrong[k] is not None and (streams[k][1] & 0x20):
            ro, _ = oracle.decompress_dict(streams[k], wrong[k])
            assert ro.status == 20 and r.value.show() == ro.message.decode(), k
    # unchanged default: Zlib.hs:68
    plain = P.decompress_many(streams[:10], ctx=gpu_ctx)
    for k, r in enumerate(plain):
        ro, oo = oracle.decompress(streams[k], len(datas[k]) + 64)
        assert (r == P.Right(oo)) if ro.status == 0 else (r.value.show() == ro.message.decode())


def test_multi_member_gzip_extension(gpu_ctx, oracle):
    """RFC 1952 2.2: a gzip file is a series of members -- decoded into one output inside one stream; every member's
    ISIZE and the combined CRC-32 are checked.  Valid files against gzip.decompress, broken ones against the oracle."""
    from test_gpu_parity import run_batch
    streams, datas = [], []
    for seed in range(80):
        parts = [corpus.mixed_data((seed * 977 + 131 * k) % 40000, seed + k) for k in range(1 + seed % 4)]
        g = b"".join(corpus.gzip_member(p, seed + k) if k % 2 else gzip.compress(p, 1 + seed % 9) for k, p in enumerate(parts))
        streams.append(g)
        datas.append(b"".join(parts))
        assert gzip.decompress(g) == datas[-1]
    for ring in (11, 15):
        gpu_ctx.set_ring_bits(ring)
        (out_len, status, detail, in_used, crc), outs, _, _ = run_batch(gpu_ctx, streams, [len(d) + 8 for d in datas], gzip=True)
        for k in range(len(streams)):
            assert status[k] == 0 and outs[k] == datas[k] and int(crc[k]) == zlib.crc32(datas[k]) and int(in_used[k]) == len(streams[k]), k
    gpu_ctx.set_ring_bits(11)
    bad = [corpus.corrupt(streams[k % 80], 7000 + k) for k in range(400)]
    caps = [len(datas[k % 80]) + 4096 for k in range(400)]
    (out_len, status, detail, in_used, crc), outs, _, _ = run_batch(gpu_ctx, bad, caps, gzip=True)
    for k in range(len(bad)):
        r, o = oracle.gzip_decompress(bad[k], caps[k])
        if int(status[k]) == 14 and r.status in (10, 19):
            continue
        assert int(status[k]) == r.status, (k, status[k], r.status, r.message)
        if r.status == 0:
            assert outs[k] == o and int(crc[k]) == r.adler
        elif r.status in (10, 19):
            assert [int(detail[k][0]), int(detail[k][1])] == [r.detail0, r.detail1], k
