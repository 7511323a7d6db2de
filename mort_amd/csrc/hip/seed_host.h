/*
 * seed_host.h -- host side of curand_init(seed, subsequence, 0) (rng.cuh:8-15 through cuRAND's XORWOW): the seed
 * scramble and the 2^67-step sequence skip as GF(2) matrices.  Used by mort_hip_rng_seed (matrices uploaded for
 * seed_kernel) and by mort_hip_rng_seed_host (`mort --mode host`, no GPU).
 */
#ifndef MORT_SEED_HOST_H
#define MORT_SEED_HOST_H

#include <stdint.h>
#include <cstring>
#include <utility>
#include <vector>

#define SEQ_LEVELS 32

/* GF(2) 160x160 matrix helpers for the sequence skip (own implementation of the
 * published XORWOW jump: state(n + 2^67 k) = M^k state(n), d unchanged). */
struct XMat { uint32_t row[160][5]; };
static inline void xmat_apply(const XMat &m, const uint32_t v[5], uint32_t out[5]) {
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < 5; w++)
        for (int b = 0; b < 32; b++)
            if ((v[w] >> b) & 1u) for (int k = 0; k < 5; k++) r[k] ^= m.row[w * 32 + b][k];
    std::memcpy(out, r, sizeof r);
}
static inline void xmat_square(const XMat &a, XMat &out) { for (int i = 0; i < 160; i++) xmat_apply(a, a.row[i], out.row[i]); }
static inline void build_seq_matrices(std::vector<XMat> &seq) {
    XMat *a = new XMat, *b = new XMat;
    for (int i = 0; i < 160; i++) {
        uint32_t v[5] = {0, 0, 0, 0, 0};
        v[i / 32] = 1u << (i % 32);
        uint32_t t = v[0] ^ (v[0] >> 2);
        uint32_t n4 = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
        a->row[i][0] = v[1]; a->row[i][1] = v[2]; a->row[i][2] = v[3]; a->row[i][3] = v[4]; a->row[i][4] = n4;
    }
    for (int k = 0; k < 67; k++) { xmat_square(*a, *b); std::swap(a, b); }
    seq.resize(SEQ_LEVELS);
    seq[0] = *a;
    for (int k = 1; k < SEQ_LEVELS; k++) { xmat_square(seq[k - 1], *b); xmat_square(*b, seq[k]); }
    delete a; delete b;
}


/* cuRAND XORWOW seed scramble (curand_kernel.h, restated from the public header: SURVEY 8c) */
struct SeedWords { uint32_t d, v[5]; };
static inline SeedWords seed_scramble(uint64_t seed) {
    const uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
    const uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    const uint32_t t0 = 1099087573u * s0;
    const uint32_t t1 = 2591861531u * s1;
    SeedWords w;
    w.d = 6615241u + t1 + t0;
    w.v[0] = 123456789u + t0; w.v[1] = 362436069u ^ t0; w.v[2] = 521288629u + t1; w.v[3] = 88675123u ^ t1; w.v[4] = 5783321u + t0;
    return w;
}
/* state of subsequence p: base-4 digits of p times A^(2^67 * 4^k), as seed_kernel does */
static inline void seed_subsequence(const std::vector<XMat> &seq, const SeedWords &w, unsigned long long p, uint32_t v[5]) {
    std::memcpy(v, w.v, sizeof w.v);
    for (int k = 0; p != 0 && k < SEQ_LEVELS; k++, p >>= 2)
        for (int t = (int)(p & 3ull); t > 0; t--) { uint32_t r[5]; xmat_apply(seq[k], v, r); std::memcpy(v, r, sizeof r); }
}

#endif
