/*
 * ORACLE (test infrastructure - never linked into, called by, or shipped with the product).
 *
 * Plain-C restatement (OpenMP over rows) of the same arithmetic as oracle/ref_np.py, i.e. of
 * CompactFusion's residual codecs:
 *   1-bit  : xfuser/compact/fastpath.py:124-228, :371-438   (scale prologue :150-166)
 *   2-bit  : xfuser/compact/fastpath.py:584-669, :745-811   (scale prologue :614-625)
 *   int8   : xfuser/compact/compress_quantize.py:428-484 on delta (+ main.py:227-233)
 *   int4   : xfuser/compact/compress_quantize.py:522-640 on delta (+ main.py:227-233)
 *   top-k  : xfuser/compact/compress_topk.py:11-163 (slowpath.py:76-79)
 * Wire layouts as in include/cfx.h.  Used (a) by tests as a second, independent checker for sizes
 * the numpy oracle is slow at, (b) by bench.py's `cpu_baseline` leg ("kind": "port") to time the
 * reference algorithm on the host cores of the GPU box - the Python reference cannot travel there.
 *
 * Parity status: PINNED - tests/test_oracle_c.py checks this file bit-for-bit against
 * oracle/ref_np.py, which is pinned against the reference's golden vectors.
 *
 * fp16 arithmetic: every op converts to fp32, operates, rounds once to fp16 (round-to-nearest-even);
 * by Figueroa's theorem that equals correctly rounded fp16 arithmetic (= torch eager, = the HIP kernels).
 * Portable C by default, so the same .so runs on whatever CPU the GPU box has.  Built a second time with -DCFX_F16C
 * -mf16c -mavx2 (libcfx_oracle_f16c.so) the two fp16 conversions use the hardware instructions (IEEE round-to-nearest-even,
 * the same results, tests/test_oracle_c.py runs both); c_oracle.py loads that build when /proc/cpuinfo lists f16c + avx2.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#ifdef CFX_F16C
#include <immintrin.h>
#endif

typedef uint16_t h16;

static float H2F[65536];
static int tables_ready = 0;

static float h2f_slow(h16 h) {
    uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31, m = h & 1023, u;
    if (e == 0) {
        if (m == 0) u = s;
        else {
            int sh = 0;
            while (!(m & 1024)) { m <<= 1; ++sh; }
            u = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((m & 1023) << 13);
        }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

void oracle_init(void) {
    if (tables_ready) return;
    for (uint32_t i = 0; i < 65536; ++i) H2F[i] = h2f_slow((h16)i);
    tables_ready = 1;
}

#ifdef CFX_F16C
static inline float h2f(h16 h) { return _cvtsh_ss(h); }
static inline h16 f2h(float f) { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }
#define f2h_soft f2h_unused
#else
static inline float h2f(h16 h) { return H2F[h]; }
#define f2h_soft f2h
#endif

/* fp32 -> fp16, round to nearest even, IEEE (subnormals, inf, nan) */
static inline h16 f2h_soft(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t s = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return (h16)(s | 0x7c00u | (u > 0x7f800000u ? 0x200u | ((u >> 13) & 0x3ffu) : 0));
    if (u >= 0x477ff000u) return (h16)(s | 0x7c00u);                 /* >= 65520 -> inf */
    if (u < 0x33000001u) return (h16)s;                              /* <= 2^-25 -> 0 (ties to even = 0) */
    if (u < 0x38800000u) {                                           /* subnormal half */
        const int e = (int)(u >> 23);
        const uint32_t m = (u & 0x7fffffu) | 0x800000u;
        const int sh = 126 - e;                                      /* 14..24 */
        const uint32_t q = m >> sh, rem = m & ((1u << sh) - 1), half = 1u << (sh - 1);
        uint32_t r = q;
        if (rem > half || (rem == half && (q & 1))) r++;
        return (h16)(s | r);
    }
    const uint32_t m = u & 0x1fffu;
    uint32_t r = (u - 0x38000000u) >> 13;
    if (m > 0x1000u || (m == 0x1000u && (r & 1))) r++;
    return (h16)(s | r);
}

static inline h16 hadd(h16 a, h16 b) { return f2h(h2f(a) + h2f(b)); }
static inline h16 hsub(h16 a, h16 b) { return f2h(h2f(a) - h2f(b)); }
static inline h16 hmul(h16 a, h16 b) { return f2h(h2f(a) * h2f(b)); }
static inline h16 hdivh(h16 a, h16 b) { return f2h(h2f(a) / h2f(b)); }
static inline h16 hrint(h16 a) { return f2h(nearbyintf(h2f(a))); }  /* default rounding mode: ties to even */
static inline int hnan(h16 a) { return (a & 0x7fffu) > 0x7c00u; }
static inline h16 hneg(h16 a) { return (h16)(a ^ 0x8000u); }

static inline uint64_t habs_units(h16 b) {
    const uint32_t e = (b >> 10) & 31u, m = b & 1023u;
    const uint32_t t = e ? (m | 1024u) : m;
    return (uint64_t)t << (e ? e - 1u : 0u);
}
static inline h16 mean16(uint64_t units, int n) { return f2h(((float)units * 0x1p-24f) / (float)n); }

enum { BINARY = 1, INT2 = 2, INT4 = 3, INT8 = 4, TOPK = 5 };
enum { F_UPDATE = 1, F_NO_EF = 2 };

size_t oracle_packet_bytes(int codec, int N, int C, int param) {
    const size_t n = N, c = C;
    switch (codec) {
        case BINARY: return n * c / 8 + 2 * (n + c);
        case INT2: return n * c / 4 + 2 * (n + c);
        case INT4: return n * c / 2 + 4 * c;
        case INT8: return n * c + 4 * c;
        case TOPK: return 2 * (n * c / param) + n * c / (2 * param);
    }
    return 0;
}

/* delta row into d[] */
static inline void delta_row(const h16* x, const h16* b, h16* d, int C) {
    if (b) for (int c = 0; c < C; ++c) d[c] = hsub(x[c], b[c]);
    else memcpy(d, x, (size_t)C * 2);
}

static void absmean_scales(const h16* x, const h16* base, int N, int C, h16* U, h16* V, int eps_mode) {
    uint64_t* col = (uint64_t*)calloc((size_t)C, 8);
    h16* um = (h16*)malloc((size_t)N * 2);
#pragma omp parallel
    {
        uint64_t* lc = (uint64_t*)calloc((size_t)C, 8);
        h16* d = (h16*)malloc((size_t)C * 2);
#pragma omp for schedule(static)
        for (int n = 0; n < N; ++n) {
            delta_row(x + (size_t)n * C, base ? base + (size_t)n * C : NULL, d, C);
            uint64_t rs = 0;
            for (int c = 0; c < C; ++c) { const uint64_t u = habs_units(d[c]); lc[c] += u; rs += u; }
            um[n] = mean16(rs, C);
        }
#pragma omp critical
        for (int c = 0; c < C; ++c) col[c] += lc[c];
        free(lc); free(d);
    }
    uint64_t tot = 0;
    for (int n = 0; n < N; ++n) tot += habs_units(um[n]);
    const h16 mu = mean16(tot, N);
    const float den = eps_mode ? h2f(f2h(h2f(mu) + 1e-6f)) : h2f(mu);
    for (int n = 0; n < N; ++n) U[n] = f2h(h2f(um[n]) / den);
    for (int c = 0; c < C; ++c) V[c] = mean16(col[c], N);
    free(col); free(um);
}

static inline h16 int2_recv(unsigned idx, h16 thr) {
    const h16 lvl = (idx & 1u) ? hmul(0x4000 /*2.0*/, thr) : hmul(0x3800 /*0.5*/, thr);
    return (idx & 2u) ? lvl : hneg(lvl);
}

static void store_state(h16* nb, const h16* x, const h16* b, const h16* recv, int C, int flags) {
    if (!(flags & F_UPDATE) || !nb) return;
    if (flags & F_NO_EF) { memcpy(nb, x, (size_t)C * 2); return; }
    for (int c = 0; c < C; ++c) nb[c] = b ? hadd(b[c], recv[c]) : recv[c];
}

int oracle_compress(int codec, const h16* x, const h16* base, h16* new_base, uint8_t* pk, int N, int C, int param, int flags) {
    oracle_init();
    if (codec == BINARY || codec == INT2) {
        const int per = codec == BINARY ? 8 : 4;
        h16* U = (h16*)(pk + (size_t)N * (C / per));
        h16* V = U + N;
        absmean_scales(x, base, N, C, U, V, codec == INT2);
#pragma omp parallel
        {
            h16* d = (h16*)malloc((size_t)C * 2);
            h16* rv = (h16*)malloc((size_t)C * 2);
#pragma omp for schedule(static)
            for (int n = 0; n < N; ++n) {
                const h16* xr = x + (size_t)n * C;
                const h16* br = base ? base + (size_t)n * C : NULL;
                delta_row(xr, br, d, C);
                uint8_t* q = pk + (size_t)n * (C / per);
                memset(q, 0, (size_t)(C / per));
                for (int c = 0; c < C; ++c) {
                    const float df = h2f(d[c]);
                    const unsigned s = df >= 0.0f;
                    if (codec == BINARY) {
                        q[c >> 3] |= (uint8_t)(s << (c & 7));
                        const h16 sc = hmul(U[n], V[c]);
                        rv[c] = s ? sc : hneg(sc);
                    } else {
                        const h16 thr = hmul(V[c], U[n]);
                        const unsigned m = h2f((h16)(d[c] & 0x7fffu)) > h2f(thr);
                        const unsigned idx = (s << 1) | m;
                        q[c >> 2] |= (uint8_t)(idx << (2 * (c & 3)));
                        rv[c] = int2_recv(idx, thr);
                    }
                }
                store_state(new_base ? new_base + (size_t)n * C : NULL, xr, br, rv, C, flags);
            }
            free(d); free(rv);
        }
        return 0;
    }
    if (codec == INT4 || codec == INT8) {
        h16* S = (h16*)(pk + (codec == INT4 ? (size_t)(N / 2) * C : (size_t)N * C));
        h16* M = S + C;                 /* int4: min ; int8: zp (int16) */
        float* mn = (float*)malloc((size_t)C * 4);
        float* mx = (float*)malloc((size_t)C * 4);
        for (int c = 0; c < C; ++c) { mn[c] = INFINITY; mx[c] = -INFINITY; }
#pragma omp parallel
        {
            float* lmn = (float*)malloc((size_t)C * 4);
            float* lmx = (float*)malloc((size_t)C * 4);
            h16* d = (h16*)malloc((size_t)C * 2);
            for (int c = 0; c < C; ++c) { lmn[c] = INFINITY; lmx[c] = -INFINITY; }
#pragma omp for schedule(static)
            for (int n = 0; n < N; ++n) {
                delta_row(x + (size_t)n * C, base ? base + (size_t)n * C : NULL, d, C);
                for (int c = 0; c < C; ++c) { const float v = h2f(d[c]); if (v < lmn[c]) lmn[c] = v; if (v > lmx[c]) lmx[c] = v; }
            }
#pragma omp critical
            for (int c = 0; c < C; ++c) { if (lmn[c] < mn[c]) mn[c] = lmn[c]; if (lmx[c] > mx[c]) mx[c] = lmx[c]; }
            free(lmn); free(lmx); free(d);
        }
        for (int c = 0; c < C; ++c) {
            const h16 hmn = f2h(mn[c]), hmx = f2h(mx[c]);
            const h16 rng = hsub(hmx, hmn);
            if (codec == INT4) { S[c] = f2h(h2f(rng) / 15.000001f); M[c] = hmn; }
            else {
                const h16 sc = f2h(h2f(rng) / 255.000001f);
                const h16 r = hrint(hdivh(hmn, sc));
                h16 z = hsub(0xd800 /* -128 */, r);
                int16_t zi;
                if (hnan(z)) zi = 0;
                else { float zf = h2f(z); if (zf < -128.f) zf = -128.f; if (zf > 127.f) zf = 127.f; zi = (int16_t)zf; }
                S[c] = sc;
                ((int16_t*)M)[c] = zi;
            }
        }
        free(mn); free(mx);
#pragma omp parallel
        {
            h16* d = (h16*)malloc((size_t)C * 2);
            h16* rv = (h16*)malloc((size_t)C * 2);
#pragma omp for schedule(static)
            for (int k = 0; k < N / (codec == INT4 ? 2 : 1); ++k) {
                const int rows = codec == INT4 ? 2 : 1;
                for (int h = 0; h < rows; ++h) {
                    const int n = k * rows + h;
                    const h16* xr = x + (size_t)n * C;
                    const h16* br = base ? base + (size_t)n * C : NULL;
                    delta_row(xr, br, d, C);
                    for (int c = 0; c < C; ++c) {
                        if (codec == INT8) {
                            const h16 zp = f2h((float)((int16_t*)M)[c]);
                            h16 v = hrint(hadd(hdivh(d[c], S[c]), zp));
                            float vf = hnan(v) ? 0.f : h2f(v);
                            if (vf < -128.f) vf = -128.f; if (vf > 127.f) vf = 127.f;
                            ((int8_t*)pk)[(size_t)n * C + c] = (int8_t)vf;
                            rv[c] = hmul(hsub(f2h(vf), zp), S[c]);
                        } else {
                            h16 v = hrint(hdivh(hsub(d[c], M[c]), S[c]));
                            float vf = hnan(v) ? 0.f : h2f(v);
                            if (vf < 0.f) vf = 0.f; if (vf > 15.f) vf = 15.f;
                            uint8_t* q = pk + (size_t)k * C + c;
                            if (h == 0) *q = (uint8_t)vf; else *q |= (uint8_t)((unsigned)vf << 4);
                            rv[c] = hadd(hmul(f2h(vf), S[c]), M[c]);
                        }
                    }
                    store_state(new_base ? new_base + (size_t)n * C : NULL, xr, br, rv, C, flags);
                }
            }
            free(d); free(rv);
        }
        return 0;
    }
    if (codec == TOPK) {
        const size_t E = (size_t)N * C;
        const int m = param;
        h16* val = (h16*)pk;
        uint8_t* idx = (uint8_t*)(val + E / m);
        const long nblk = (long)(E / (2 * m));
#pragma omp parallel for schedule(static)
        for (long b = 0; b < nblk; ++b) {
            h16 d[32], rv[32];
            const size_t e0 = (size_t)b * 2 * m;
            for (int i = 0; i < 2 * m; ++i) { d[i] = base ? hsub(x[e0 + i], base[e0 + i]) : x[e0 + i]; rv[i] = 0; }
            unsigned sel[2];
            for (int h = 0; h < 2; ++h) {
                int best = 0;
                float bv = h2f((h16)(d[h * m] & 0x7fffu));
                for (int i = 1; i < m; ++i) { const float v = h2f((h16)(d[h * m + i] & 0x7fffu)); if (v > bv) { bv = v; best = i; } }
                sel[h] = (unsigned)best;
                val[b * 2 + h] = d[h * m + best];
                rv[h * m + best] = d[h * m + best];
            }
            idx[b] = (uint8_t)((sel[0] << 4) | sel[1]);
            if ((flags & F_UPDATE) && new_base) {
                for (int i = 0; i < 2 * m; ++i)
                    new_base[e0 + i] = (flags & F_NO_EF) ? x[e0 + i] : (base ? hadd(base[e0 + i], rv[i]) : rv[i]);
            }
        }
        return 0;
    }
    return -4;
}

int oracle_decompress(int codec, const uint8_t* pk, const h16* base, h16* out, int N, int C, int param) {
    oracle_init();
    if (codec == BINARY || codec == INT2) {
        const int per = codec == BINARY ? 8 : 4;
        const h16* U = (const h16*)(pk + (size_t)N * (C / per));
        const h16* V = U + N;
#pragma omp parallel for schedule(static)
        for (int n = 0; n < N; ++n) {
            const uint8_t* q = pk + (size_t)n * (C / per);
            for (int c = 0; c < C; ++c) {
                h16 rv;
                if (codec == BINARY) {
                    const h16 sc = hmul(U[n], V[c]);
                    rv = ((q[c >> 3] >> (c & 7)) & 1) ? sc : hneg(sc);
                } else {
                    rv = int2_recv((q[c >> 2] >> (2 * (c & 3))) & 3u, hmul(V[c], U[n]));
                }
                out[(size_t)n * C + c] = base ? hadd(base[(size_t)n * C + c], rv) : rv;
            }
        }
        return 0;
    }
    if (codec == INT4 || codec == INT8) {
        const h16* S = (const h16*)(pk + (codec == INT4 ? (size_t)(N / 2) * C : (size_t)N * C));
        const h16* M = S + C;
#pragma omp parallel for schedule(static)
        for (int n = 0; n < N; ++n) {
            for (int c = 0; c < C; ++c) {
                h16 rv;
                if (codec == INT8) {
                    const h16 zp = f2h((float)((const int16_t*)M)[c]);
                    rv = hmul(hsub(f2h((float)((const int8_t*)pk)[(size_t)n * C + c]), zp), S[c]);
                } else {
                    const unsigned q = (pk[(size_t)(n >> 1) * C + c] >> (4 * (n & 1))) & 15u;
                    rv = hadd(hmul(f2h((float)q), S[c]), M[c]);
                }
                out[(size_t)n * C + c] = base ? hadd(base[(size_t)n * C + c], rv) : rv;
            }
        }
        return 0;
    }
    if (codec == TOPK) {
        const size_t E = (size_t)N * C;
        const int m = param;
        const h16* val = (const h16*)pk;
        const uint8_t* idx = (const uint8_t*)(val + E / m);
#pragma omp parallel for schedule(static)
        for (long e = 0; e < (long)E; ++e) {
            const size_t hb = (size_t)e / m;
            const unsigned by = idx[hb >> 1];
            const unsigned sel = (hb & 1) ? (by & 15u) : (by >> 4);
            const h16 rv = ((unsigned)(e % m) == sel) ? val[hb] : 0;
            out[e] = base ? hadd(base[e], rv) : rv;
        }
        return 0;
    }
    return -4;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_uses_f16c(void) {
#ifdef CFX_F16C
    return 1;
#else
    return 0;
#endif
}
