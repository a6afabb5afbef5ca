// device_math.h -- the per-point / per-row arithmetic of the path, shared by both engines.
//
// Everything here must round exactly like the reference's SSE2 build: fp32 throughout,
// no fused multiply-add (the translation units are compiled with -ffp-contract=off),
// IEEE division, round-half-even where the reference uses _mm_cvtps_epi32.
// Reference paths are under /root/reference/Thirdparty/DenseCRF/include/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lccrf {

// ---------------------------------------------------------------------------------------
// lattice coordinates of one point.  ref: permutohedral_cpu.h:304-366
//   out: r0[i]  = i-th coordinate of the enclosing simplex's remainder-0 vertex (i < D)
//        rk[i]  = rank of coordinate i                                          (i < D)
//        b[rem] = barycentric weight of the simplex corner with remainder `rem` (rem <= D)
// ---------------------------------------------------------------------------------------
// the enclosing simplex: el = the elevated point, rem0 = its remainder-0 vertex (all D+1 coordinates, as floats), rank = the rank of
// every coordinate after the off-plane fix-up.  ref: permutohedral_cpu.h:304-345
template <int D>
__device__ __forceinline__ void lattice_simplex(const float (&feat)[D], const float *scale, float inv_dp1, float (&el)[D + 1],
                                                float (&rem0)[D + 1], float (&rank)[D + 1])
{
    constexpr int D1 = D + 1;
    const float dp1 = (float)D1;

    // elevate, :304-310
    float sm = 0.0f;
#pragma unroll
    for (int j = D; j > 0; --j) {
        const float cf = feat[j - 1] * scale[j - 1];
        el[j] = sm - (float)j * cf;
        sm += cf;
    }
    el[0] = sm;

    // nearest remainder-0 point; _mm_cvtps_epi32 rounds half to even, :313-323 (quirk Q2)
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < D1; ++i) {
        const float v = (float)__float2int_rn(inv_dp1 * el[i]);
        rem0[i] = v * dp1;
        sum += v;
        rank[i] = 0.0f;
    }

    // rank by strict fp32 '<', :326-336 (quirk Q3)
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const float di = el[i] - rem0[i];
#pragma unroll
        for (int j = i + 1; j < D1; ++j) {
            const float dj = el[j] - rem0[j];
            const float c = (di < dj) ? 1.0f : 0.0f;
            rank[i] += c;
            rank[j] += 1.0f - c;
        }
    }

    // off-plane fix-up, :339-345
#pragma unroll
    for (int i = 0; i < D1; ++i) {
        rank[i] += sum;
        const float add = (rank[i] < 0.0f) ? dp1 : 0.0f;
        const float sub = (rank[i] >= dp1) ? dp1 : 0.0f;
        const float adj = add - sub;
        rank[i] += adj;
        rem0[i] += adj;
    }
}

template <int D>
__device__ __forceinline__ void point_record(const float (&feat)[D], const float *scale, float inv_dp1,
                                             int16_t (&r0)[D], uint8_t (&rk)[D], float (&b)[D + 1])
{
    constexpr int D1 = D + 1;
    float el[D1], rem0[D1], rank[D1];
    lattice_simplex<D>(feat, scale, inv_dp1, el, rem0, rank);

    // barycentric weights, accumulated over i = 0..D in order, :348-366.
    // bb has D+2 cells; the indexed updates are written as selects so it stays in registers.
    float bb[D1 + 1];
#pragma unroll
    for (int q = 0; q < D1 + 1; ++q) bb[q] = 0.0f;
#pragma unroll
    for (int i = 0; i < D1; ++i) {
        const float v = (el[i] - rem0[i]) * inv_dp1;
        const int p = (int)((float)D - rank[i]);
#pragma unroll
        for (int q = 0; q < D1 + 1; ++q) {
            if (q == p) bb[q] = bb[q] + v;
            if (q == p + 1) bb[q] = bb[q] - v;
        }
    }
    bb[0] += 1.0f + bb[D1];

#pragma unroll
    for (int i = 0; i < D; ++i) {
        r0[i] = (int16_t)(int)rem0[i];
        rk[i] = (uint8_t)(int)rank[i];
    }
#pragma unroll
    for (int i = 0; i < D1; ++i) b[i] = bb[i];
}

// point_record<2> for the grid build of frame_lean.hip -- the same simplex, the same barycentric BITS, other outputs:
//   b[rem]       barycentric weights.  The reference accumulates b[p] += v_i, b[p + 1] -= v_i over i = 0..2 with p = 2 - rank_i
//                (:348-366); the ranks are a permutation of {0, 1, 2}, so cell q receives exactly +v of the coordinate with p = q and
//                -v of the one with p = q - 1, in either order: (0 + a) - c and (0 - c) + a are the same fp32 value for every finite
//                a, c (signed zeros included: both give +0 when a = c = +-0), so sorting the three v by p and differencing gives the
//                reference's bits with a quarter of the compare / select instructions of the indexed form.
//   r0x, r0y     first two coordinates of the remainder-0 vertex (multiples of 3), as ints
//   f            bit 0: rank_x > 1, bit 1: rank_x > 0, bit 2: rank_y > 1, bit 3: rank_y > 0 -- corner `rem` of the simplex has the key
//                (r0x + rem - 3 [rank_x > 2 - rem], r0y + rem - 3 [rank_y > 2 - rem])  (vertex_coord<2>; rem = 0: the vertex itself)
//   wrapped      set when a coordinate is beyond +-32000: the reference's int16 keys may wrap there (not the grid build's case)
__device__ __forceinline__ void point_record2_grid(const float (&feat)[2], const float *scale, float inv_dp1, float (&b)[3], int &r0x,
                                                   int &r0y, unsigned &f, bool &wrapped)
{
    float el[3], rem0[3], rank[3];
    lattice_simplex<2>(feat, scale, inv_dp1, el, rem0, rank);
    float v[3], vq[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = (el[i] - rem0[i]) * inv_dp1;
#pragma unroll
    for (int q = 0; q < 3; ++q) {                        // vq[q] = the v whose p = 2 - rank is q
        vq[q] = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i) vq[q] = (rank[i] == (float)(2 - q)) ? v[i] : vq[q];
    }
    const float b3 = 0.0f - vq[2];
    b[0] = (0.0f + vq[0]) + (1.0f + b3);                 // b[0] += 1 + b[D + 1], :366
    b[1] = (0.0f + vq[1]) - vq[0];
    b[2] = (0.0f + vq[2]) - vq[1];
    r0x = (int)rem0[0];
    r0y = (int)rem0[1];
    f = (rank[0] > 1.0f ? 1u : 0u) | (rank[0] > 0.0f ? 2u : 0u) | (rank[1] > 1.0f ? 4u : 0u) | (rank[1] > 0.0f ? 8u : 0u);
    wrapped |= !(fabsf(rem0[0]) < 32000.0f) || !(fabsf(rem0[1]) < 32000.0f);
}

// i-th key coordinate of the corner with remainder `rem`: rem0 + canonical[rem][rank],
// canonical[rem][r] = rem if r <= D-rem else rem-(D+1).  ref: :274-279,373.
template <int D>
__device__ __forceinline__ int16_t vertex_coord(int16_t r0, uint8_t rk, int rem)
{
    const int c = ((int)rk <= D - rem) ? rem : rem - (D + 1);
    return (int16_t)(r0 + c);
}

// Any hash works: vertex numbering comes from first-occurrence order, not from the table.
template <int D>
__device__ __forceinline__ unsigned hash_key(const int16_t (&key)[D])
{
    unsigned h = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        h += (unsigned)(int)key[i];
        h *= 1664525u;
    }
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    return h;
}

// ---------------------------------------------------------------------------------------
// softmax with the reference's polynomial exp.  ref: densecrf3d.h:51-98
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float very_fast_exp(float x)     // densecrf3d.h:51-54
{
    return 1 - x * (0.9999999995f - x * (0.4999999206f - x * (0.1666653019f - x * (0.0416573475f
             - x * (0.0083013598f - x * (0.0013298820f - x * (0.0001413161f)))))));
}

// The reference compares the float against DOUBLE products (0.69*2*2*2 etc., quirk Q5).
// None of the three is representable in fp32 and each rounds DOWN when narrowed, so no fp32
// value lies between (float)T and T:  (double)x > T  <=>  x > (float)T.  Checked at compile time.
constexpr double kFe8d = 0.69 * 2 * 2 * 2, kFe4d = 0.69 * 2 * 2, kFe1d = 0.69;
constexpr float kFe8 = (float)kFe8d, kFe4 = (float)kFe4d, kFe1 = (float)kFe1d;
static_assert((double)kFe8 < kFe8d && (double)kFe4 < kFe4d && (double)kFe1 < kFe1d,
              "fast_exp thresholds must narrow downwards for the fp32 compare to be equivalent");

__device__ __forceinline__ float fast_exp(float x)          // densecrf3d.h:55-67
{
    // x == 0 (the row maximum in a softmax): the reference takes its 1/x branch with
    // very_fast_exp(0) = 1 and returns 1/1.  Same value, without the division.
    if (x == 0) return 1.0f;
    bool less_zero = true;
    if (x < 0) { less_zero = false; x = -x; }
    if (x > 20) return 0;
    int mult = 0;
    while (x > kFe8) { mult += 3; x /= 8.0f; }
    while (x > kFe4) { mult += 2; x /= 4.0f; }
    while (x > kFe1) { mult += 1; x /= 2.0f; }
    x = very_fast_exp(x);
    while (mult) { mult--; x = x * x; }
    return less_zero ? 1 / x : x;
}

// fast_exp for arguments <= 0 (all a softmax ever passes: value - row maximum), with the
// data-dependent loops of densecrf3d.h:60-64 turned into selects so a wavefront does not
// serialise over its lanes' different trip counts.  Same operations, same order, same
// results -- including x == 0 (-> 1) and x < -20 (-> 0):
//   * a = -x <= 20 enters the "/8" loop at most once (20/8 < 5.52), the "/4" loop at most
//     once, the "/2" loop at most twice; dividing by 8, 4, 2 is exact, as is multiplying
//     by 0.125, 0.25, 0.5;
//   * the reference then squares `mult` (<= 5) times.
__device__ __forceinline__ float fast_exp_nonpos(float x)
{
    // The cascade "a > 5.52 ? a/8, a > 2.76 ? a/4, a > 0.69 ? a/2, a > 0.69 ? a/2" only ever scales by exact powers of two
    // and kFe4 == 4 kFe1, kFe8 == 8 kFe1 exactly (rounding commutes with scaling by 2^n), so every one of its compares is a
    // compare of the ORIGINAL a against a power-of-two multiple of kFe1, and for a <= 20 the number of halvings is
    //     mult = [a > kFe1] + [a > 2 kFe1] + [a > 4 kFe1] + [a > 8 kFe1] + [a > 16 kFe1]
    // (0.69 | 1.38 | 2.76 | 5.52 | 11.04: e.g. 5.52 < a <= 11.04 takes /8, not /4 since a/8 <= 1.38, then one /2),
    // the reduced argument a * 2^-mult: five compares and one v_ldexp_f32 instead of four compare-select-multiply rounds.
    static_assert(kFe4 == 4.0f * kFe1 && kFe8 == 8.0f * kFe1, "thresholds are power-of-two multiples of one another");
    float a = -x;
    const bool cut = a > 20;
#ifndef LCCRF_EXP_FREXP
#define LCCRF_EXP_FREXP 1                 // A/B (scripts/gpu_ab_build.sh "" "-DLCCRF_EXP_FREXP=0"): 0 = the five compares
#endif
#if LCCRF_EXP_FREXP
    // ... and those five compares are a frexp: kFe1 = 0.69f is its own mantissa (in [0.5, 1), exponent 0), a = m * 2^e with m in
    // [0.5, 1), so  a > kFe1 * 2^i  <=>  e > i or (e == i and m > kFe1)  and the count over i = 0 .. 4 is clamp(e + [m > kFe1], 0, 5)
    // (a = 0: e = 0, m = 0 -> 0; a < 0.5: e <= -1 -> 0; a >= 16: e >= 5 -> 5).  Two frexp instructions, a compare, an add, a clamp
    // instead of five compare / select / add rounds: the X phase is VALU-bound once two frames share a CU (round 5).
    static_assert(kFe1 >= 0.5f && kFe1 < 1.0f, "kFe1 must be its own frexp mantissa");
    const int e2 = __builtin_amdgcn_frexp_expf(a);
    const float m2 = __builtin_amdgcn_frexp_mantf(a);
    const int mult = min(max(e2 + (m2 > kFe1 ? 1 : 0), 0), 5);
#else
    const int mult = (a > kFe1 ? 1 : 0) + (a > 2.0f * kFe1 ? 1 : 0) + (a > kFe4 ? 1 : 0) + (a > kFe8 ? 1 : 0) + (a > 16.0f * kFe1 ? 1 : 0);
#endif
    a = __builtin_amdgcn_ldexpf(a, -mult);
    float r = very_fast_exp(a);
#pragma unroll
    for (int i = 0; i < 5; ++i) r = (i < mult) ? r * r : r;
    return cut ? 0.0f : r;
}

// One row of expAndNormalize (densecrf3d.h:70-98); `out` may alias `in`.
__device__ __forceinline__ void exp_and_normalize_row(const float *in, float *out, int L, float scale,
                                                      float relax)
{
    float mx = scale * in[0];
    for (int j = 1; j < L; ++j) {
        const float s = scale * in[j];
        if (mx < s) mx = s;
    }
    float tt = 0;
    for (int j = 0; j < L; ++j) tt += fast_exp(scale * in[j] - mx);
    for (int j = 0; j < L; ++j) {
        const float v = fast_exp(scale * in[j] - mx) / tt;
        if (relax == 1) out[j] = v;
        else out[j] = (1 - relax) * out[j] + relax * v;
    }
}

// Compile-time-width variant for values already in registers.
template <int L>
__device__ __forceinline__ void exp_and_normalize_reg(const float (&in)[L], float (&out)[L], float scale,
                                                      float relax)
{
    float mx = scale * in[0];
#pragma unroll
    for (int j = 1; j < L; ++j) {
        const float s = scale * in[j];
        if (mx < s) mx = s;
    }
    float v[L];
    float tt = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) { v[j] = fast_exp_nonpos(scale * in[j] - mx); tt += v[j]; }
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const float p = v[j] / tt;
        if (relax == 1) out[j] = p;
        else out[j] = (1 - relax) * out[j] + relax * p;
    }
}

// expAndNormalize for two labels (densecrf3d.h:70-98), a = scale*in[0], b = scale*in[1].  One of
// the two fast_exp arguments is exactly 0 (value minus row maximum) and fast_exp(0) == 1, so a
// single exp is evaluated; the sum and the two IEEE divisions are the reference's.
// omr = 1 - relax, formed once by the caller (the same fp32 subtraction as densecrf3d.h:94, (1 - relax): a uniform value the
// kernels would otherwise keep in a vector register for the whole launch)
#ifndef LCCRF_QUOT_STEPS
#define LCCRF_QUOT_STEPS 2                // A/B (scripts/gpu_ab_build.sh "" "-DLCCRF_QUOT_STEPS=1"): residual corrections per quotient
#endif
__device__ __forceinline__ float2 softmax2_fresh(float a, float b)
{
    const bool lt = a < b;                            // mx = b iff a < b (densecrf3d.h:76-79)
    const float e = fast_exp_nonpos(lt ? a - b : b - a);
    const float tt = 1.0f + e;                        // v0 + v1 with one of them exactly 1 (fp32 addition commutes)
    // The two IEEE divisions 1/tt and e/tt, written out: tt is in [1, 2] and e in {0} U [2^-29, 1], a range in which
    // hipcc's own expansion of x/y (v_div_scale, v_rcp, Newton step, quotient + two residual corrections, v_div_fmas,
    // v_div_fixup) scales nothing and fixes nothing up -- what is left is this sequence, with the refined reciprocal
    // shared by both quotients (13 instructions instead of 22; same bits, tests compare Q bit for bit).
    // (Round 6: ONE residual correction per quotient would do -- both quotients are functions of e alone, and
    // scripts/ubench/quotcheck.hip finds 0 differences from IEEE division over EVERY float e in {0} U [2^-60, 1] -- but the
    // four instructions it saves per point bought nothing measurable (C2 +-0, C4 / N500 +0.4 %) and cost two frame-kernel
    // variants 8-12 bytes of scratch: LCCRF_QUOT_STEPS stays 2.)
    const float r0 = __builtin_amdgcn_rcpf(tt);
    const float r = __builtin_fmaf(__builtin_fmaf(-tt, r0, 1.0f), r0, r0);
    auto quot = [&](float n) {
        const float q = n * r;
#if LCCRF_QUOT_STEPS == 2
        const float q2 = __builtin_fmaf(__builtin_fmaf(-tt, q, n), r, q);
        return __builtin_fmaf(__builtin_fmaf(-tt, q2, n), r, q2);
#else
        return __builtin_fmaf(__builtin_fmaf(-tt, q, n), r, q);
#endif
    };
    const float pm = quot(1.0f), pe = quot(e);
    return make_float2(lt ? pe : pm, lt ? pm : pe);
}
// the lean loops' form: the damping (densecrf3d.h:91-94) behind a real, uniform branch -- left to itself the compiler if-converts it
// and relax = 1, every caller of the reference, pays the blend's five instructions per point (C2 inference: 1.2 %)
__device__ __forceinline__ float2 softmax2(float a, float b, float2 old, float relax, float omr)
{
    float2 out = softmax2_fresh(a, b);
    if (__builtin_expect(relax != 1, 0)) {
        asm volatile("");
        out = make_float2(omr * old.x + relax * out.x, omr * old.y + relax * out.y);
    }
    return out;
}
__device__ __forceinline__ float2 softmax2(float a, float b, float2 old, float relax)
{
    const float2 p = softmax2_fresh(a, b);
    if (relax == 1) return p;
    const float omr = 1 - relax;
    return make_float2(omr * old.x + relax * p.x, omr * old.y + relax * p.y);
}

__device__ __forceinline__ int argmax_row(const float *p, int L)   // densecrf3d.h:140-149
{
    float mx = p[0];
    int imx = 0;
    for (int m = 1; m < L; ++m)
        if (mx < p[m]) { mx = p[m]; imx = m; }
    return imx;
}

}  // namespace lccrf
