/*
 * pz_baseline_mt.c -- TEST INFRASTRUCTURE: the multi-threaded CPU baselines bench.py times beside the GPU path
 * (never linked into the product).  Two decoders over the product ABI's batch layout, on nthreads POSIX threads
 * drawing stream indices from one atomic counter:
 *   pzo_decompress_many_mt   the oracle (oracle/pz_oracle.c: the bit-at-a-time restatement of pure-zlib)
 *   pzo_zlib_many_mt         system zlib's uncompress() (libz 1.2.11), the specification-equal C decoder
 * Both return the number of streams that failed or whose length differs from out_cap[i]; *bytes gets the total decoded.
 * adler_out (may be NULL): per stream, the Adler-32 of the bytes THIS decoder produced (the oracle's own running checksum,
 * Adler32.hs:17-57; for zlib, adler32() over its output) -- bench.py compares it with the GPU's adler[] for every stream
 * of the timed batch (GPU vs restatement on every stream, not a sample).
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <zlib.h>

#include "pz_oracle.h"

typedef struct {
    const uint8_t *in_base;
    const uint64_t *in_off, *in_len, *out_cap;
    uint32_t n;
    uint32_t *adler_out;
    int use_zlib;
    atomic_uint next;
    atomic_ullong bytes;
    atomic_uint bad;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    uint64_t cap = 0, bytes = 0;
    uint8_t *out = NULL;
    uint32_t bad = 0;
    for (;;) {
        uint32_t i = atomic_fetch_add(&j->next, 16u), e = i + 16u;
        if (i >= j->n) break;
        if (e > j->n) e = j->n;
        for (; i < e; i++) {
            if (j->out_cap[i] + 64 > cap) {
                free(out);
                cap = j->out_cap[i] + 64;
                out = (uint8_t *)malloc(cap);
            }
            if (j->use_zlib) {
                uLongf dl = (uLongf)cap;
                if (uncompress(out, &dl, j->in_base + j->in_off[i], (uLong)j->in_len[i]) != Z_OK || dl != j->out_cap[i]) {
                    bad++;
                    if (j->adler_out) j->adler_out[i] = 0u;
                } else {
                    bytes += dl;
                    if (j->adler_out) j->adler_out[i] = (uint32_t)adler32(1L, out, (uInt)dl);
                }
            } else {
                pzo_result r;
                pzo_decompress(j->in_base + j->in_off[i], j->in_len[i], out, j->out_cap[i], &r);
                if (r.status != PZO_OK || r.out_len != j->out_cap[i]) bad++;
                else bytes += r.out_len;
                if (j->adler_out) j->adler_out[i] = r.adler;
            }
        }
    }
    free(out);
    atomic_fetch_add(&j->bytes, bytes);
    atomic_fetch_add(&j->bad, bad);
    return NULL;
}

static uint32_t run(const uint8_t *in_base, const uint64_t *in_off, const uint64_t *in_len, const uint64_t *out_cap, uint32_t n,
                    uint32_t nthreads, int use_zlib, uint64_t *bytes, uint32_t *adler_out)
{
    job_t j;
    pthread_t *th;
    uint32_t t, started = 0;
    j.in_base = in_base;
    j.in_off = in_off;
    j.in_len = in_len;
    j.out_cap = out_cap;
    j.n = n;
    j.adler_out = adler_out;
    j.use_zlib = use_zlib;
    atomic_init(&j.next, 0u);
    atomic_init(&j.bytes, 0ull);
    atomic_init(&j.bad, 0u);
    if (nthreads < 1) nthreads = 1;
    th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    for (t = 1; t < nthreads; t++)
        if (pthread_create(&th[started], NULL, worker, &j) == 0) started++;
    worker(&j);
    for (t = 0; t < started; t++) pthread_join(th[t], NULL);
    free(th);
    if (bytes) *bytes = atomic_load(&j.bytes);
    return atomic_load(&j.bad);
}

uint32_t pzo_decompress_many_mt(const uint8_t *in_base, const uint64_t *in_off, const uint64_t *in_len, const uint64_t *out_cap,
                                uint32_t n, uint32_t nthreads, uint64_t *bytes, uint32_t *adler_out)
{
    return run(in_base, in_off, in_len, out_cap, n, nthreads, 0, bytes, adler_out);
}

uint32_t pzo_zlib_many_mt(const uint8_t *in_base, const uint64_t *in_off, const uint64_t *in_len, const uint64_t *out_cap, uint32_t n,
                          uint32_t nthreads, uint64_t *bytes, uint32_t *adler_out)
{
    return run(in_base, in_off, in_len, out_cap, n, nthreads, 1, bytes, adler_out);
}
