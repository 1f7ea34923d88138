/*
 * af_noise.h — the engine's counter-based noise source ("tier B" generator).
 *
 * The reference (genData/player.py:240,270,275,278,102,125) draws its Dirichlet
 * noise and tie-breaks from the process-global, serial MT19937 streams of
 * numpy and Python `random`.  A serial stream cannot be reproduced by thousands
 * of concurrent games on a GPU, so the engine DEFINES its own generator here:
 * Philox4x32-10 keyed by (seed, game id) and indexed by
 * (select counter, episode, stream|cell, iteration).  The sampling ALGORITHMS
 * are the reference's (numpy legacy gamma-rejection for Dirichlet(alpha<1),
 * uniform index draws, inverse-cdf move sampling); only the uniform source and
 * the log/exp implementation (and, for the Dirichlet noise, their fp32 precision) are the build's own.
 *
 * Everything in this header is written with plain IEEE-754 +,-,*,/ (no fma, no
 * libm) so that gcc on the host and hipcc on gfx950 produce bit-identical
 * results when both compile with -ffp-contract=off and without fast-math.
 * It is included by the HIP engine (product) and by oracle/af_oracle.c (the
 * checker) — it is a specification shared by both, like a file format.
 */
#ifndef AF_NOISE_H
#define AF_NOISE_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define AF_HD __host__ __device__ static inline
#else
#define AF_HD static inline
#endif

/* ---- noise streams (third counter word = stream << 24 | index) ---- */
#define AF_STREAM_GAMMA   0u  /* index = cell, iteration = rejection round        */
#define AF_STREAM_PICK    1u  /* uniform pick among select candidates             */
#define AF_STREAM_BEST    2u  /* per-ply tie-break among most-visited (random.choice) */
#define AF_STREAM_MOVE    3u  /* per-ply move sample (np.random.choice(L,p))      */

typedef struct { uint32_t v[4]; } af_u32x4;

AF_HD uint32_t af_mulhi32(uint32_t a, uint32_t b) {
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}

/* Philox4x32-10 (Salmon et al., SC'11).  ctr[4], key[2] -> 4 words. */
AF_HD af_u32x4 af_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                             uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        /* (one 64-bit product per multiplier: on the GPU a single v_mad_u64_u32 instead of v_mul_hi_u32 + v_mul_lo_u32) */
        const uint64_t p0 = (uint64_t)M0 * (uint64_t)c0, p1 = (uint64_t)M1 * (uint64_t)c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    af_u32x4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

/* 53-bit uniform in [0,1) from two words — same construction as MT19937's
 * genrand_res53 used by numpy's legacy double. */
AF_HD double af_u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

/* per-game key */
AF_HD uint32_t af_key0(uint64_t seed) { return (uint32_t)seed; }
AF_HD uint32_t af_key1(uint64_t seed, uint32_t game_id) { return (uint32_t)(seed >> 32) + game_id; }

/* ---- deterministic log / exp (plain-op, ~1e-15 relative) ---- */
AF_HD double af_bits2d(uint64_t u) { union { uint64_t u; double d; } x; x.u = u; return x.d; }
AF_HD uint64_t af_d2bits(double d) { union { uint64_t u; double d; } x; x.d = d; return x.u; }

#define AF_NEG_HUGE (-1.0e300)

/* natural log of a positive finite double; af_log(0) = AF_NEG_HUGE */
AF_HD double af_log(double x) {
    if (!(x > 0.0)) return AF_NEG_HUGE;
    uint64_t b = af_d2bits(x);
    int e = (int)(b >> 52);
    if (e == 0) {                        /* subnormal: scale up by 2^54 */
        x = x * 18014398509481984.0;
        b = af_d2bits(x);
        e = (int)(b >> 52) - 54;
    }
    e -= 1023;
    uint64_t mant = b & 0x000FFFFFFFFFFFFFull;
    double m = af_bits2d(mant | 0x3FF0000000000000ull);   /* [1,2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }   /* [0.7071,1.4142] */
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (0.3999999999940941908 + w * (0.2222219843214978396 + w * 0.1531383769920937332));
    double t2 = z * (0.6666666666666735130 + w * (0.2857142874366239149 + w * (0.1818357216161805012 + w * 0.1479819860511658591)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    double dk = (double)e;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

/* exp(x); returns 0 below ~-708, saturates above +709 */
AF_HD double af_exp(double x) {
    if (x < -708.0) return 0.0;
    if (x > 709.0) x = 709.0;
    const double inv_ln2 = 1.44269504088896338700e+00;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    double kf = x * inv_ln2;
    int k = (int)(kf + (kf < 0.0 ? -0.5 : 0.5));
    double dk = (double)k;
    double hi = x - dk * ln2_hi;
    double lo = dk * ln2_lo;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * (1.66666666666666019037e-01 + t * (-2.77777777770155933842e-03 + t * (6.61375632143793436117e-05 +
               t * (-1.65339022054652515390e-06 + t * 4.13813679705723846039e-08))));
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    /* scale by 2^k in two exact steps (k in [-1022, 1023]) */
    int k1 = k / 2, k2 = k - k1;
    double s1 = af_bits2d((uint64_t)(k1 + 1023) << 52);
    double s2 = af_bits2d((uint64_t)(k2 + 1023) << 52);
    return (y * s1) * s2;
}

/* x^y for x >= 0 (x == 0 -> 0), via exp(y*log x) */
AF_HD double af_pow(double x, double y) {
    if (!(x > 0.0)) return 0.0;
    return af_exp(y * af_log(x));
}

/* fp32 power used by the tier-B temperature policy (player.py:117) */
AF_HD float af_powf(float x, float y) {
    if (!(x > 0.0f)) return 0.0f;
    return (float)af_exp((double)y * af_log((double)x));
}

/* ---- samplers ---- */

/* ---- fp32 plain-op log / exp: the arithmetic of the gamma sampler ----
 * The Dirichlet noise only perturbs priors (player.py:247-253), so the sampler runs in fp32: half the polynomial
 * length of the fp64 pair above and no fp64 division — it is ~70 % of a select's instructions on the GPU.
 * FreeBSD msun e_logf.c / e_expf.c restated with plain +,-,*,/ (no fma): < 2 ulp, bit-identical under gcc and hipcc. */
AF_HD float af_bits2f(uint32_t u) { union { uint32_t u; float f; } x; x.u = u; return x.f; }
AF_HD uint32_t af_f2bits(float f) { union { uint32_t u; float f; } x; x.f = f; return x.u; }

#define AF_NEG_HUGE_F (-1.0e30f)

/* natural log of a positive normal float; af_logf(x <= 0) = AF_NEG_HUGE_F (arguments here are >= 2^-24).
 * Straight-line code (r4): the x <= 0 case is a select at the end, not an early return — on the GPU an early return is a divergent
 * branch (exec-mask save / restore) in front of every log of every rejection round, and it fences the two independent logs of a
 * round (V and Y) off from each other; the arithmetic and so every result bit is unchanged. */
AF_HD float af_logf(float x) {
    const uint32_t b = af_f2bits(x);
    int e = (int)(b >> 23) - 127;
    float m = af_bits2f((b & 0x007FFFFFu) | 0x3F800000u);      /* [1,2) */
    const int up = m > 1.41421356f;                             /* -> [0.7071,1.4142] */
    m = up ? m * 0.5f : m;
    e += up;
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float w = z * z;
    const float t1 = w * (0.40000972152f + w * 0.24279078841f);
    const float t2 = z * (0.66666662693f + w * 0.28498786688f);
    const float R = t2 + t1;
    const float hfsq = 0.5f * f * f;
    const float dk = (float)e;
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float r = dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    return (x > 0.0f) ? r : AF_NEG_HUGE_F;
}

/* exp(x); 0 below -87, saturates above 88 (straight-line like af_logf: the argument is clamped, the underflow case selected at the end) */
AF_HD float af_expf(float x) {
    const float xc = x > 88.0f ? 88.0f : (x < -87.0f ? -87.0f : x);
    const float inv_ln2 = 1.4426950216e+00f, ln2_hi = 6.9314575195e-01f, ln2_lo = 1.4286067653e-06f;
    const float kf = xc * inv_ln2;
    const int k = (int)(kf + (kf < 0.0f ? -0.5f : 0.5f));
    const float dk = (float)k;
    const float hi = xc - dk * ln2_hi, lo = dk * ln2_lo;
    const float r = hi - lo;
    const float t = r * r;
    const float c = r - t * (1.6666625440e-01f + t * -2.7667332906e-03f);
    const float y = 1.0f - ((lo - (r * c) / (2.0f - c)) - hi);
    const float v = y * af_bits2f((uint32_t)(k + 127) << 23);   /* k in [-126, 127] */
    return x < -87.0f ? 0.0f : v;
}

/* 24-bit uniform in [0, 1 - 2^-24] from one word */
AF_HD float af_u24(uint32_t a) { return (float)(a >> 8) * 5.9604644775390625e-08f; }

/* One round of numpy legacy_standard_gamma's rejection loop for alpha < 1 (restated in SURVEY.md §8a) on two
 * words (U, and V = Exp(1)); returns 1 and *x when the candidate is accepted.  Written without data-dependent
 * branches so that a 64-lane wavefront runs every round in lockstep: both arms share one pow, the second arm's log is
 * evaluated for all (its argument (1-U)/alpha is >= 1 in the first arm) and V + 0 == V in the first arm. */
AF_HD int af_gamma_round(float alpha, float inv_a, float one_m_a, uint32_t wu, uint32_t wv, float* x) {
    const float U = af_u24(wu);
    const float V = -af_logf(1.0f - af_u24(wv));
    const float Yl = -af_logf((1.0f - U) * inv_a);
    const int first = U <= one_m_a;
    const float Y = first ? 0.0f : Yl;
    const float base = first ? U : one_m_a + alpha * Yl;
    const float X = af_expf(inv_a * af_logf(base));             /* base == 0 -> exp(-huge) = 0 */
    *x = X;
    return X <= V + Y;
}

/* One Gamma(alpha,1) variate, alpha < 1: rounds it = 0,1,... until one accepts.  Round `it` takes words
 * 2*(it&1), 2*(it&1)+1 of Philox(counter = (sel, episode, GAMMA<<24|cell, it>>1)): one block feeds two rounds. */
AF_HD double af_gamma_lt1(double alpha, uint32_t sel, uint32_t episode, uint32_t cell,
                          uint32_t k0, uint32_t k1) {
    const float a = (float)alpha, inv_a = 1.0f / a, one_m_a = 1.0f - a;
    af_u32x4 r;
    for (uint32_t it = 0;; ++it) {
        if ((it & 1u) == 0u) r = af_philox4x32(sel, episode, (AF_STREAM_GAMMA << 24) | cell, it >> 1, k0, k1);
        float X;
        if (af_gamma_round(a, inv_a, one_m_a, r.v[2 * (it & 1u)], r.v[2 * (it & 1u) + 1], &X)) return (double)X;
        if (it == 0xFFFFu) return 0.0;   /* unreachable in practice; bounds the loop */
    }
}

/* Dirichlet(alpha * 1_L) over the legal cells = the gamma variates above divided by their sum (fp64; summation tree:
 * oracle/af_oracle.c:afo_noise_philox_dirichlet).  If EVERY variate is 0 (fp32 underflow of exp(log(U)/alpha): small alpha,
 * few legal cells) the draw is defined as the uniform distribution over the legal cells instead of 0/0. */

/* uniform index in [0,m): multiply-shift on one word; m == 1 -> 0 */
AF_HD uint32_t af_pick(uint32_t m, uint32_t sel, uint32_t episode, uint32_t stream,
                       uint32_t k0, uint32_t k1) {
    if (m <= 1u) return 0u;
    af_u32x4 r = af_philox4x32(sel, episode, stream << 24, 0u, k0, k1);
    return af_mulhi32(r.v[0], m);
}

/* one uniform double for the per-ply move sample */
AF_HD double af_uniform(uint32_t sel, uint32_t episode, uint32_t stream, uint32_t k0, uint32_t k1) {
    af_u32x4 r = af_philox4x32(sel, episode, stream << 24, 0u, k0, k1);
    return af_u53(r.v[0], r.v[1]);
}

#endif /* AF_NOISE_H */
