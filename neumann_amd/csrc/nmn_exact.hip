// nmn_exact.hip — kernels that restate the reference's f32 arithmetic BIT FOR BIT on the GPU.
// THIS TRANSLATION UNIT IS BUILT WITH -ffp-contract=off (build.py) and carries a file-scope
// `#pragma clang fp contract(off)`: hipcc contracts a*b+c into v_fma by default, and HIP's
// __fmul_rn/__fadd_rn are plain operators that contraction would fuse.  (The only FMAs left in its ISA
// are inside the correctly rounded division / square-root expansions.)  A contracted build cannot pass
// the bit-level parity tests in tests/test_gpu_parity_basic.py.
//
// Reference order (tensor_store/src/hnsw.rs:168-229, vector_engine/src/lib.rs:2231-2266):
//   dot8 / sumsq8 : 8 accumulators, acc[l] = acc[l] + (a[8c+l]*b[8c+l]) for c = 0..d/8-1 (mul and
//                   add rounded separately), r = ((((((((-0+acc0)+acc1)+...)+acc7), then the scalar
//                   tail r = r + a[i]*b[i].
//   euclidean     : strictly sequential s = s + (x-y)*(x-y), sqrt.
//   cosine        : dot / (|q| * |v|), 0 if either magnitude is 0;  euclid score 1/(1+dist).
// Work split: 8 consecutive threads own one (query,row) pair, thread l runs accumulator lane l's
// dependent chain; the 8 partial sums are then added left to right by every thread of the group
// (shuffles inside the 8-lane group), so no reassociation ever happens.
#include <algorithm>
#include <cstring>

#include "nmn_select_dev.h"

// Belt and braces: even if a build forgets -ffp-contract=off, nothing below may be contracted.
#pragma clang fp contract(off)

namespace nmn {

__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
// IEEE-correct under -fhip-fp32-correctly-rounded-divide-sqrt (passed explicitly by build.py).
// NOT HIP's __fsqrt_rn / __fdiv_rn: without OCML_BASIC_ROUNDED_OPERATIONS __fsqrt_rn is the
// *native* (approximate) square root.
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }
__device__ __forceinline__ float sqrt_rn(float a) { return __builtin_sqrtf(a); }

// lane-l chain of dot8 followed by the in-order lane sum and the scalar tail.
// `l` = threadIdx & 7; all 8 threads of the group return the same value.  Loads are issued PF chunks
// ahead of the dependent mul/add chain (the chain order is untouched; only the memory latency of the
// 8 x (d/8) strided reads is overlapped).
template <int PF>
__device__ __forceinline__ float dot8_group(const float* __restrict__ a, const float* __restrict__ b,
                                            uint32_t dim, uint32_t l) {
    const uint32_t chunks = dim >> 3;
    float acc = 0.0f;
    for (uint32_t c0 = 0; c0 < chunks; c0 += PF) {
        float av[PF], bv[PF];
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const uint32_t c = c0 + (uint32_t)i;
            av[i] = c < chunks ? a[8u * c + l] : 0.0f;
            bv[i] = c < chunks ? b[8u * c + l] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < PF; i++) {
            if (c0 + (uint32_t)i < chunks) {
                const float pr = mul_rn(av[i], bv[i]);
                acc = add_rn(acc, pr);
            }
        }
    }
    float r = -0.0f;
    const int base = (int)(threadIdx.x & 63u & ~7u);
#pragma unroll
    for (int t = 0; t < 8; t++) r = add_rn(r, __shfl(acc, base + t));
    for (uint32_t i = chunks * 8u; i < dim; i++) r = add_rn(r, mul_rn(a[i], b[i]));
    return r;
}

// lane T of the caller's 8-lane group (lanes 8g .. 8g+7 of the wave), to all eight: two DPP moves — row_newbcast:T into the
// lanes 0-7 of every 16-lane DPP row (bank_mask 0x3), row_newbcast:8+T into its lanes 8-15 (bank_mask 0xC) — instead of a
// ds_bpermute through the LDS crossbar (what __shfl with a computed lane compiles to: ~70 cycles a piece in a dependent chain)
template <int T>
__device__ __forceinline__ float group8_bcast(float v) {
    int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + T, 0xF, 0x3, false);
    r = __builtin_amdgcn_update_dpp(r, __float_as_int(v), 0x158 + T, 0xF, 0xC, false);
    return __int_as_float(r);
}

// lib.rs:2249-2253: one strictly sequential sum.  The 8 threads of a group split the LOADS and the
// (x-y)^2 products (thread l owns elements i with i % 8 == l); the sum itself is then accumulated in
// element order by pulling each product from its owner (group8_bcast), so the order of the additions is
// exactly 0,1,2,...,d-1.  Elements past `dim` (the last chunk of a row whose length is not a multiple of eight, the chunks a
// batch of PF reaches beyond the row) contribute (0 - 0)^2 = +0.0: added to a sum that is +0.0 or larger once the first real
// product is in — the running sum starts at -0.0 and -0.0 + (+0.0) = +0.0, every product being >= +0.0 — they change nothing,
// so the additions carry no bounds test (the test cost the chain a compare and a select per element).
// Round 4: 58 -> see profiles/r04k_* us for the 6 400 rows x 1536 elements a TOP-1000 query's crowd re-scores.
template <int PF>
__device__ __forceinline__ float euclid_sumsq_seq(const float* __restrict__ q, const float* __restrict__ v,
                                                  uint32_t dim, uint32_t l) {
    float s = -0.0f;
    const uint32_t chunks = (dim + 7u) >> 3;
    // the loads of batch b + 1 are in flight under the additions of batch b (the chain does not depend on them): a batch used to
    // wait for its own loads first — twelve round trips per 1536-element row, two thirds of the row's time
    float xa[PF], ya[PF];
    auto load = [&](uint32_t c0, float (&x)[PF], float (&y)[PF]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const uint32_t e = 8u * (c0 + (uint32_t)i) + l;
            x[i] = e < dim ? q[e] : 0.0f;
            y[i] = e < dim ? v[e] : 0.0f;
        }
    };
    load(0, xa, ya);
    for (uint32_t c0 = 0; c0 < chunks; c0 += PF) {
        float xn[PF], yn[PF];
        load(c0 + PF, xn, yn);  // (past the end: every element fails e < dim, nothing is read)
        float pr[PF];
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const float d = sub_rn(xa[i], ya[i]);
            pr[i] = mul_rn(d, d);
        }
#pragma unroll
        for (int i = 0; i < PF; i++) {
            s = add_rn(s, group8_bcast<0>(pr[i]));
            s = add_rn(s, group8_bcast<1>(pr[i]));
            s = add_rn(s, group8_bcast<2>(pr[i]));
            s = add_rn(s, group8_bcast<3>(pr[i]));
            s = add_rn(s, group8_bcast<4>(pr[i]));
            s = add_rn(s, group8_bcast<5>(pr[i]));
            s = add_rn(s, group8_bcast<6>(pr[i]));
            s = add_rn(s, group8_bcast<7>(pr[i]));
        }
#pragma unroll
        for (int i = 0; i < PF; i++) {
            xa[i] = xn[i];
            ya[i] = yn[i];
        }
    }
    return s;
}

// tensor_blob's artifact similarity (tensor_blob/src/lib.rs:601-603): both vectors go through
// SparseVector::from_dense (keeps every value != 0.0, NaN included; sparse_vector.rs:221-229) and
// SparseVector::cosine_similarity (583-599): dot_f64 = sequential f64 sum of f64(a_i)*f64(b_i) over the positions
// where BOTH are stored (419-443), magnitude_f64 = sqrt of the sequential f64 sum of squares of the stored values
// (553-559), result = dot / (mag_a * mag_b) with 0.0 for a zero magnitude or a NaN/Inf result, clamped to
// [-1, 1] and rounded once to f32.  Products of two f32 are exact in f64, and a skipped position contributes
// exactly what adding +0.0 does (the running sums start at 0.0 and can never become -0.0), so the 8 threads of
// the group form the products and every thread then adds them in index order.
template <int PF>
__device__ __forceinline__ float sparse_cos64(const float* __restrict__ q, const float* __restrict__ v, uint32_t dim,
                                              uint32_t l) {
    double dot = 0.0, sa = 0.0, sb = 0.0;
    const int base = (int)(threadIdx.x & 63u & ~7u);
    const uint32_t chunks = (dim + 7u) >> 3;
    for (uint32_t c0 = 0; c0 < chunks; c0 += PF) {
        double pd[PF], pa[PF], pb[PF];
#pragma unroll
        for (int i = 0; i < PF; i++) {
            const uint32_t e = 8u * (c0 + (uint32_t)i) + l;
            const float x = e < dim ? q[e] : 0.0f;
            const float y = e < dim ? v[e] : 0.0f;
            const bool sx = x != 0.0f, sy = y != 0.0f;  // stored in the sparse form (true for NaN)
            pd[i] = (sx && sy) ? (double)x * (double)y : 0.0;
            pa[i] = sx ? (double)x * (double)x : 0.0;
            pb[i] = sy ? (double)y * (double)y : 0.0;
        }
#pragma unroll
        for (int i = 0; i < PF; i++) {
#pragma unroll
            for (int t = 0; t < 8; t++) {
                dot = dot + __shfl(pd[i], base + t);
                sa = sa + __shfl(pa[i], base + t);
                sb = sb + __shfl(pb[i], base + t);
            }
        }
    }
    const double mag_a = __builtin_sqrt(sa), mag_b = __builtin_sqrt(sb);
    if (mag_a == 0.0 || mag_b == 0.0) return 0.0f;
    const double r = dot / (mag_a * mag_b);
    if (r != r || __builtin_isinf(r)) return 0.0f;
    const double c = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
    return (float)c;
}

// compute_score (lib.rs:2231-2266) for one (query,row); vmag = stored simd::magnitude(row).
// The two internal metrics serve the IVF probe (tensor_store/src/ivf.rs:500-508 `squared_euclidean` is the same
// sequential sum as euclidean_distance's): kMetricNegL2Sq ranks centroids by squared distance (ivf.rs:331-337),
// kMetricNegL2 ranks list members by `squared_euclidean(..).sqrt()` (ivf.rs:365-369); negated so that
// "nearest first" is the descending order every later stage works in (negation is exact).
__device__ __forceinline__ float exact_score(const float* __restrict__ q, const float* __restrict__ v,
                                             uint32_t dim, float qmag, float vmag, int metric, uint32_t l) {
    if (metric == NMN_METRIC_EUCLIDEAN || metric == kMetricNegL2 || metric == kMetricNegL2Sq) {
        const float ss = euclid_sumsq_seq<16>(q, v, dim, l);
        if (metric == kMetricNegL2Sq) return -ss;
        const float dist = sqrt_rn(ss);
        if (metric == kMetricNegL2) return -dist;
        return div_rn(1.0f, add_rn(1.0f, dist));
    }
    if (metric == NMN_METRIC_SPARSE_COSINE_F64) return sparse_cos64<8>(q, v, dim, l);
    const float dot = dot8_group<32>(q, v, dim, l);
    if (metric == NMN_METRIC_DOT_PRODUCT) return dot;
    if (qmag == 0.0f || vmag == 0.0f) return 0.0f;
    return div_rn(dot, mul_rn(qmag, vmag));
}

// ---- |v| for uploaded rows -------------------------------------------------------------------
__global__ void __launch_bounds__(256) norms_kernel(const float* __restrict__ corpus, uint32_t ld, uint32_t dim,
                                                    uint64_t row0, uint64_t n, float* __restrict__ norms,
                                                    float* __restrict__ inv_norms, uint32_t* __restrict__ max_norm_bits) {
    const uint32_t l = threadIdx.x & 7u;
    const uint64_t i = (uint64_t)blockIdx.x * 32u + (threadIdx.x >> 3);
    const uint64_t row = row0 + (i < n ? i : n - 1);  // keep the whole 8-group converged for the shuffles
    const float* v = corpus + row * (uint64_t)ld;
    const float ss = dot8_group<32>(v, v, dim, l);
    const float mag = sqrt_rn(ss);
    if (i < n && l == 0) {
        norms[row] = mag;
        inv_norms[row] = mag == 0.0f ? 0.0f : div_rn(1.0f, mag);  // approximate sweeps only: cosine = dot * (1/|q|) * (1/|v|)
        if (mag == mag) atomicMax(max_norm_bits, f2u(mag));  // mag >= 0: bit order == value order
    }
}

hipError_t launch_norms(const float* corpus, uint32_t ld, uint32_t dim, uint64_t row0, uint64_t n, float* norms,
                        float* inv_norms, uint32_t* max_norm_bits, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + 31) / 32;
    hipLaunchKernelGGL(norms_kernel, dim3((unsigned)blocks), dim3(256), 0, s, corpus, ld, dim, row0, n, norms, inv_norms,
                       max_norm_bits);
    return hipGetLastError();
}

// ---- query preparation -----------------------------------------------------------------------
// One block of 64 threads per query: zero-padded copy, |q| in reference order, error margins of the
// approximate scan, reset of the per-query selection state.
__global__ void __launch_bounds__(64) qprep_kernel(const float* __restrict__ queries, uint32_t dim, uint32_t ld,
                                                   int metric, const uint32_t* __restrict__ max_norm_bits,
                                                   float* __restrict__ qpad, QInfo* __restrict__ qinfo,
                                                   QState* __restrict__ qstate, int mfma_pass,
                                                   const uint32_t* __restrict__ half_err_bits, uint32_t* __restrict__ qi8,
                                                   const float* __restrict__ l2_hint, QInfo* __restrict__ qinfo_plain,
                                                   uint32_t* __restrict__ run_slots, uint32_t run_S, uint32_t* __restrict__ run_bound) {
    const uint32_t q = blockIdx.x;
    if (run_S) {  // the one-launch batched sweep derives its score-store bound from these (ScanParams::run_*)
        for (uint32_t i = threadIdx.x; i < 1024u; i += 64u) run_slots[(size_t)q * 1024u + i] = kKeyMasked;  // (ScanParams::run_slots: [nq][1024])
        if (threadIdx.x == 0) run_bound[q] = kKeyNaN;
    }
    // The query comes into LDS in ONE round trip (every load of the block in flight together) and every later phase — the padded
    // copy, |q| in reference order, the bf16 / int8 roundings — reads it there.  (Until round 5 each phase looped over global
    // memory, a dependent load per 64 elements: 10 us at 768 elements and 25 us at 1536 in front of EVERY search, profiles/r05g_*.)
    extern __shared__ __attribute__((aligned(16))) float qprep_lds[];  // [ld]
    float* const src = qprep_lds;
    {
        const float* gsrc = queries + (size_t)q * dim;
        if ((dim & 3u) == 0u && (reinterpret_cast<uintptr_t>(gsrc) & 15u) == 0u) {
            typedef float qf4 __attribute__((ext_vector_type(4)));
            const uint32_t n4 = dim >> 2;
            for (uint32_t i0 = threadIdx.x; i0 < n4; i0 += 64u * 8u) {
                qf4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t i = i0 + (uint32_t)u * 64u;
                    v[u] = i < n4 ? reinterpret_cast<const qf4*>(gsrc)[i] : (qf4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t i = i0 + (uint32_t)u * 64u;
                    if (i < n4) reinterpret_cast<qf4*>(src)[i] = v[u];
                }
            }
        } else {
            for (uint32_t i0 = threadIdx.x; i0 < dim; i0 += 64u * 16u) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const uint32_t i = i0 + (uint32_t)u * 64u;
                    v[u] = i < dim ? gsrc[i] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const uint32_t i = i0 + (uint32_t)u * 64u;
                    if (i < dim) src[i] = v[u];
                }
            }
        }
        for (uint32_t i = dim + threadIdx.x; i < ld; i += 64) src[i] = 0.0f;
        __syncthreads();
    }
    typedef float qv4 __attribute__((ext_vector_type(4)));
    {
        qv4* dst4 = reinterpret_cast<qv4*>(qpad + (size_t)q * ld);  // (ld is a multiple of 8, qpad rows 16-byte aligned)
        for (uint32_t i = threadIdx.x; i < (ld >> 2); i += 64) dst4[i] = reinterpret_cast<const qv4*>(src)[i];
    }
    const float ss = dot8_group<32>(src, src, dim, threadIdx.x & 7u);
    // MFMA sweep: the stationary copy of this query is bf16 — its rounding error |q - bf16(q)| goes into the margin
    float qerr2 = 0.0f;
    float qscale = 0.0f, qq8 = 0.0f;
    if ((mfma_pass & 4) && qi8) {
        // 8-bit sweep (nmn_scan_i8.hip): q = s_q (h + l / 256) + e_q with int8 vectors h, l; the planes go to qi8[q][2][ld]
        // (zero beyond dim), |e_q| into the margin exactly like the bf16 rounding of the matrix-core sweep's queries
        float mx = 0.0f;
        bool bad = false;
        for (uint32_t i = threadIdx.x; i < dim; i += 64) {
            const float ax = __builtin_fabsf(src[i]);
            bad = bad || !(ax <= 3.0e38f);
            mx = __builtin_fmaxf(mx, ax);
        }
        for (int off = 32; off > 0; off >>= 1) {
            mx = __builtin_fmaxf(mx, __shfl_xor(mx, off));
            bad = bad || (__shfl_xor((int)bad, off) != 0);
        }
        qscale = (bad || mx == 0.0f) ? 0.0f : mx / 127.0f;
        const float inv = qscale > 0.0f ? 127.0f / mx : 0.0f;
        uint32_t* hp = qi8 + (size_t)q * 2u * (ld >> 2);  // four codes per word (ld is a multiple of 8)
        uint32_t* lp = hp + (ld >> 2);
        for (uint32_t i4 = threadIdx.x; i4 < (ld >> 2); i4 += 64) {
            uint32_t hw = 0u, lw = 0u;
#pragma unroll
            for (uint32_t c = 0; c < 4u; c++) {
                const uint32_t i = i4 * 4u + c;
                float h = 0.0f, l = 0.0f;
                if (i < dim && qscale > 0.0f) {
                    const float t = src[i] * inv;
                    h = __builtin_fminf(__builtin_fmaxf(__builtin_rintf(t), -127.0f), 127.0f);
                    l = (mfma_pass & 16) ? 0.0f : __builtin_fminf(__builtin_fmaxf(__builtin_rintf((t - h) * 256.0f), -127.0f), 127.0f);  // (16: one plane)
                    const float qt = qscale * (h + l * 0.00390625f);
                    const float e = src[i] - qt;
                    qerr2 = qerr2 + e * e;
                    qq8 = qq8 + qt * qt;
                } else if (i < dim) {
                    qerr2 = bad ? __builtin_inff() : qerr2;  // a non-finite query: infinite margin, the exact paths answer
                }
                hw |= ((uint32_t)(int)h & 0xFFu) << (8u * c);
                lw |= ((uint32_t)(int)l & 0xFFu) << (8u * c);
            }
            hp[i4] = hw;
            lp[i4] = lw;
        }
        for (int off = 32; off > 0; off >>= 1) {
            qerr2 = qerr2 + __shfl_down(qerr2, off);
            qq8 = qq8 + __shfl_down(qq8, off);
        }
    } else if (mfma_pass & 1) {
        for (uint32_t i = threadIdx.x; i < dim; i += 64) {
            const float x = src[i];
            const uint32_t b = f2u(x);
            const float h = u2f((b + 0x7FFFu + ((b >> 16) & 1u)) & 0xFFFF0000u);  // round to nearest even (finite x)
            const float e = x - h;
            qerr2 = qerr2 + e * e;
        }
        for (int off = 32; off > 0; off >>= 1) qerr2 = qerr2 + __shfl_down(qerr2, off);
    }
    if (threadIdx.x == 0) {
        const float qmag = sqrt_rn(ss);
        const float u = 5.9604645e-08f;  // 2^-24
        const float dd = (float)dim;
        QInfo qi;
        qi.qmag = qmag;
        qi.pad = 0.f;
        qi.qscale = qscale;
        qi.qq8 = qq8;
        qi.pad_sq = 0.f;
        qi.neg_d = 0.f;
        // What the approximate sweep adds on top of f32 summation error, relative to |q||v|.  The margin is applied ONCE
        // below the k-th approximate score and must cover the error twice (k rows with approx >= T have exact >= T - e, so
        // the exact k-th is >= T - e, and a row with exact >= T - e has approx >= T - 2e).
        //   bf16 mirror: row r is stored as v + e_r, so |dot error| = |q . e_r| <= |q||e_r| <= |q||v_r| * rho with
        //   rho = max_r |e_r| / |v_r| MEASURED when the mirror was written (<= 2^-8, typically 0.4 * 2^-8) -> 2 rho.
        float split = 0.0f, half_abs = 0.0f;
        const float rho_v = (mfma_pass & 2) ? (half_err_bits ? u2f(half_err_bits[1]) : 3.95e-03f) : 0.0f;  // worst case 2^-8
        if (mfma_pass & 2) {  // bf16 mirror of the CORPUS (VALU and MFMA sweeps)
            split += 2.0f * rho_v;
            half_abs = half_err_bits ? 2.0f * u2f(half_err_bits[0]) : 7.9e-03f * u2f(*max_norm_bits);  // 2 max|e_r|
            // The a-priori bound |e_r| <= 2^-8 |v_r| (f32 rows rounded to bf16 in registers, no mirror whose error was measured) holds
            // for elements inside the bf16 range only: |x| > 3.39e38 rounds to +-inf, the approximate score becomes inf or NaN and the
            // row could be dropped.  Such an element makes its row's magnitude — hence the shard's largest — at least
            // that large: then no margin is claimed at all (infinite: every query of the pass takes the exact path).
            if (!half_err_bits && !(u2f(*max_norm_bits) < 3.38e38f)) {
                split = __builtin_inff();
                half_abs = __builtin_inff();
            }
        }
        if (mfma_pass & 1) {
            // bf16 QUERY on the MFMA sweep: q~ = q + e_q, v~ = v + e_v: |q~.v~ - q.v| <= |e_q||v~| + |q||e_v|
            //   <= |q||v| (rho_q (1 + rho_v) + rho_v) with rho_q = |e_q| / |q| measured just above (<= 2^-8)
            const float rho_q = qmag > 0.0f ? __builtin_sqrtf(qerr2) * 1.0005f / qmag : 0.0f;
            split += 2.0f * rho_q * (1.0f + rho_v) * 1.0005f;
        }
        if (metric == NMN_METRIC_COSINE || metric == NMN_METRIC_SPARSE_COSINE_F64) {
            qi.margin_abs = 3.0f * (dd + 10.0f) * u + split;
            qi.margin_rel = 0.0f;
        } else if (metric == NMN_METRIC_DOT_PRODUCT) {
            const float mx = u2f(*max_norm_bits);
            qi.margin_abs = (3.0f * (dd + 10.0f) * u + split) * qmag * mx;
            qi.margin_rel = 8.0f * u;
        } else {
            qi.margin_abs = 0.0f;
            qi.margin_rel = 4.0f * (dd + 8.0f) * u;
            // Which Euclidean estimator the 8-bit sweep of 1-2 queries uses.  A: |q~ - v~| between the stored representations, off by
            // <= |e_q| + |e_r| in DISTANCE space (below).  B: |q|^2 + |v|^2 - 2 q~.v~ with the exact magnitudes, off by <= 2 |q| E in
            // SQUARED-distance space (the matrix-core branch further down).  Around the threshold distance d_T, B's slack is worth
            // |q| E / d_T of distance against A's E: B is tighter exactly when d_T > |q| — uncorrelated rows (d ~ sqrt(2) |q|), where it
            // keeps 10M x 1536 TOP-1000 under a 0.1 bitmap at ~6 000 candidates instead of ~10 000 (past the selection's compact lists:
            // 0.39 ms of select_kernel) — and far looser for the near neighbours of clustered data.  Both are rigorous; the choice only
            // moves the candidate count, so it is made from the shard's PREVIOUS Euclidean selection (select_kernel leaves its
            // threshold distance in *l2_hint; 0 until one has run: estimator A).
            const bool est_b = (mfma_pass & 8) && l2_hint && *l2_hint > qmag && metric == NMN_METRIC_EUCLIDEAN;
            if (est_b) qi.qq8 = -1.0f;
            if ((mfma_pass & 4) && !est_b && (metric == NMN_METRIC_EUCLIDEAN || metric == kMetricNegL2)) {
                // 8-bit sweep: d~ = |q~ - v~| computed from the stored representations themselves, so |d~ - d| <= |e_q| + |e_r|
                // (triangle inequality): an ABSOLUTE error of the DISTANCE, twice as always -> pad.  What the f32 evaluation of
                // |q~|^2 + |v~|^2 - 2 q~.v~ adds (|q~|^2 summed by 64 lanes: (d/64 + 8) u; |v~|^2 = s^2 * an exact integer: 3 u; the
                // dot product s_q s_r (h.c + l.c / 256): 4 u; the two additions: 2 u of the larger operand) is bounded in
                // squared-distance space, also twice -> pad_sq; V~ <= V + E.
                const float V = u2f(*max_norm_bits);
                const float E = half_err_bits ? u2f(half_err_bits[0]) : 3.95e-03f * V;
                const float eq = __builtin_sqrtf(qerr2) * 1.0005f;
                const float Vt = V + E, qt = qmag + eq;
                qi.pad = 2.0f * 1.001f * (E + eq);
                qi.pad_sq = 2.0f * ((dd * 0.015625f + 16.0f) * u) * (qt + Vt) * (qt + Vt);
                qi.margin_rel = 16.0f * u;  // v_sqrt, v_rcp and the arithmetic of margin_key itself
                qi.neg_d = metric == kMetricNegL2 ? 1.0f : 0.0f;
            } else if ((mfma_pass & 1) && (metric == NMN_METRIC_EUCLIDEAN || metric == kMetricNegL2)) {
                // Matrix-core sweep: d~^2 = |q|^2 + |v|^2 - 2 q~.v~ with bf16 q~ = q + e_q, v~ = v + e_r.  Absolute
                // error of d~^2 (A):
                //   2 |q~.v~ - q.v|  <= 2 (|q||e_r| + |e_q||v| + |e_q||e_r|) <= 2 |q| (E + rho_q (V + E))
                //                       E = max_r |e_r| and rho_q = |e_q|/|q| measured, V = max_r |v_r|;
                //   f32 accumulation of the products (any order): 2 * 3 (d + 10) u |q| V  (the bound the cosine / dot
                //   matrix-core margins use); |v|^2 and |q|^2 from reference-order magnitudes: (d + 10) u (V^2 + |q|^2);
                //   the three roundings of the final expression: 3 u (|q| + V)^2.
                // Applied twice (see above) in squared-distance space by margin_key: pad = -2 A.
                const float V = u2f(*max_norm_bits);
                const float E = half_err_bits ? u2f(half_err_bits[0]) : 3.95e-03f * V;
                const float rho_q = qmag > 0.0f ? __builtin_sqrtf(qerr2) * 1.0005f / qmag : 0.0f;
                const float a_round = 2.0f * qmag * (E + rho_q * (V + E));
                const float a_fp = (dd + 10.0f) * u * (6.0f * qmag * V + V * V + qmag * qmag) + 3.0f * u * (qmag + V) * (qmag + V);
                qi.pad = -2.0f * 1.001f * (a_round + a_fp);
                qi.margin_rel = 16.0f * u;  // v_sqrt, v_rcp and the arithmetic of margin_key itself
                if (metric == kMetricNegL2) qi.margin_abs = -1.0f;  // flag for margin_key: the score is -d, not 1/(1+d)
            } else if (mfma_pass & 2) {
                // bf16 mirror under a Euclidean metric: v~ = v + e_r, so by the triangle inequality
                // |d(q, v~) - d(q, v)| <= |e_r| <= max_r |e_r| =: D, an ABSOLUTE error on the distance (twice, as above)
                const float two_d = half_abs;
                if (metric == kMetricNegL2) qi.margin_abs = two_d;  // score = -d
                else qi.pad = two_d;  // score = 1/(1+d): threshold T -> T / (1 + 2D T), applied by margin_key
            }
        }
        qinfo[q] = qi;
        if (qinfo_plain) {
            // the same query for a sweep over the f32 corpus itself (the retry of a mirror pass whose margin overflowed): what
            // this kernel writes with mfma_pass == 0 — f32 summation error only — in the same launch
            QInfo qp;
            qp.qmag = qmag;
            qp.pad = 0.f;
            qp.qscale = 0.f;
            qp.qq8 = 0.f;
            qp.pad_sq = 0.f;
            qp.neg_d = 0.f;
            if (metric == NMN_METRIC_COSINE || metric == NMN_METRIC_SPARSE_COSINE_F64) {
                qp.margin_abs = 3.0f * (dd + 10.0f) * u;
                qp.margin_rel = 0.0f;
            } else if (metric == NMN_METRIC_DOT_PRODUCT) {
                qp.margin_abs = (3.0f * (dd + 10.0f) * u) * qmag * u2f(*max_norm_bits);
                qp.margin_rel = 8.0f * u;
            } else {
                qp.margin_abs = 0.0f;
                qp.margin_rel = 4.0f * (dd + 8.0f) * u;
            }
            qinfo_plain[q] = qp;
        }
        QState st;
        st.cand_count = 0;
        st.overflow = 0;
        st.n_valid = 0;
        st.thr_key = 0;
        qstate[q] = st;
    }
}

hipError_t launch_qprep(const float* queries, uint32_t nq, uint32_t dim, uint32_t ld, int metric,
                        const uint32_t* max_norm_bits, float* qpad, QInfo* qinfo, QState* qstate, int mfma_pass,
                        hipStream_t s, const uint32_t* half_err_bits, uint32_t* qi8, const float* l2_hint, QInfo* qinfo_plain,
                        uint32_t* run_slots, uint32_t run_S, uint32_t* run_bound) {
    const size_t lds = (size_t)ld * sizeof(float);  // the query (nmn_index_create: one query fits the 160 KiB LDS)
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qprep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(qprep_kernel, dim3(nq), dim3(64), lds, s, queries, dim, ld, metric, max_norm_bits, qpad,
                       qinfo, qstate, mfma_pass, half_err_bits, qi8, l2_hint, qinfo_plain, run_slots, run_S, run_bound);
    return hipGetLastError();
}

// ---- exact rescore of the candidate lists -----------------------------------------------------
// Normal duty: re-score the <= cand_cap candidates of each query (32 per workgroup).
// Fallback duty (query flagged `overflow` by select_kernel): the same launch instead computes the exact
// score of EVERY row into scores[] (grid-stride, 32 rows per workgroup step); final_kernel then selects
// from those.  Folding the fallback into this launch keeps the normal pipeline free of extra launches
// and of any host round trip.
__global__ void __launch_bounds__(256) rescore_kernel(RescoreParams p) {
    const uint32_t q = blockIdx.y;
    const uint32_t l = threadIdx.x & 7u;
    const float* qv = p.qpad + (size_t)q * p.ld;
    const float qmag = p.qinfo[q].qmag;
    if (p.qstate[q].overflow == 2) {
        // crowd duty: exact score of every row of the query's slice (32 per workgroup step)
        const uint32_t count = p.qstate[q].cand_count;
        const uint32_t* rows = p.crowd_rows + p.crowd_offset[q];
        float* out = p.crowd_scores + p.crowd_offset[q];
        for (uint32_t c0 = blockIdx.x * 32u; c0 < count; c0 += gridDim.x * 32u) {
            const uint32_t c = c0 + (threadIdx.x >> 3);
            const uint32_t row = rows[c < count ? c : count - 1];  // keep every 8-lane group converged
            const float vmag = p.metric == NMN_METRIC_COSINE ? p.norms[row] : 1.0f;
            const float sc = exact_score(qv, p.corpus + (uint64_t)row * p.ld, p.dim, qmag, vmag, p.metric, l);
            if (c < count && l == 0) out[c] = sc;
        }
        return;
    }
    if (p.qstate[q].overflow) {
        if (p.skip_fallback) return;  // (exact_rows_kernel, a lane per row, serves this pass: nmn_ingest.hip)
        const uint64_t* mask = p.qmasks ? p.qmasks[q] : p.mask;
        const uint64_t n_pad = (p.n_rows + 63) & ~63ull;
        for (uint64_t base = (uint64_t)blockIdx.x * 32u; base < n_pad; base += (uint64_t)gridDim.x * 32u) {
            const uint64_t row = base + (threadIdx.x >> 3);
            bool valid = row < p.n_rows;
            if (valid && mask) valid = ((mask[row >> 6] >> (row & 63)) & 1ull) != 0;
            uint32_t bits = kScoreSentinelBits;
            if (valid) {  // uniform per 8-lane group
                const float vmag = p.metric == NMN_METRIC_COSINE ? p.norms[row] : 1.0f;
                bits = f2u(exact_score(qv, p.corpus + row * (uint64_t)p.ld, p.dim, qmag, vmag, p.metric, l));
            }
            if (l == 0 && row < n_pad) p.scores[score_at(row, q, p.nql)] = bits;
        }
        return;
    }
    const uint32_t count = min(p.qstate[q].cand_count, p.cand_cap);
    for (uint32_t c0 = blockIdx.x * 32u; c0 < count; c0 += gridDim.x * 32u) {
        const uint32_t c = c0 + (threadIdx.x >> 3);
        const uint32_t cc = c < count ? c : count - 1;  // keep every 8-lane group converged
        const uint32_t row = p.cand_rows[(size_t)q * p.cand_cap + cc];
        const float* v = p.corpus + (uint64_t)row * p.ld;
        const float vmag = p.metric == NMN_METRIC_COSINE ? p.norms[row] : 1.0f;
        const float sc = exact_score(qv, v, p.dim, qmag, vmag, p.metric, l);
        if (c < count && l == 0) p.cand_scores[(size_t)q * p.cand_cap + c] = sc;
    }
}

hipError_t launch_rescore(const RescoreParams& p, hipStream_t s) {
    // 4096 candidates = one step of 128 workgroups.  The fallback duty grid-strides over all rows of the
    // shard, so a launch with few queries still gets >= 2048 workgroups (the spare ones of the normal duty
    // find c0 >= count and leave at once).
    const uint32_t gx = std::max<uint32_t>(128u, (2048u + p.nq - 1) / p.nq);
    dim3 grid(gx, p.nq);
    hipLaunchKernelGGL(rescore_kernel, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---- rescore + final in ONE launch (the short chain of the host-buffer API; round 6) ------------------------------------------
// A search that waits for its answer (nmn_index_search and friends: search_enqueue's short chain) ran qprep -> sweep -> select ->
// rescore -> final: two launches at the end of a DEPENDENT chain for ~600 candidates, each paying its launch (~4.5 us between two
// dependent kernels, profiles/r05j_search_launch_chains.txt).  Here the workgroups that re-score a query's candidates take a ticket when
// they are done, and the LAST one to finish orders the (exact key, row) composites and emits the top-k — final_kernel's ordinary
// duty (sort_and_emit; a radix pre-pick when the list is longer than 1024), on 512 threads.  Release / acquire
// through agent-scope fences around the ticket, as tiny_search_kernel does.  A query whose candidate list overflowed is reported
// (out_counts[q] = 0xFFFFFFFF), never answered here: on the short chain the host follows up with the whole chain.
// `ticket` [nq]: zero between launches (the last workgroup resets its query's).
constexpr int kTailThreads = 512;  // two waves per SIMD: exact_score keeps its 200+ registers (1024 threads: 128, and it spilled — 33 us for ~600 candidates)
__global__ void __launch_bounds__(kTailThreads) rescore_final_kernel(RescoreParams p, FinalParams f, uint32_t* ticket) {
    __shared__ unsigned long long list[NMN_MAX_TOP_K];
    __shared__ uint32_t hist[kBins];
    __shared__ PickResult pick;
    __shared__ uint32_t s_misc[2];
    const uint32_t q = blockIdx.y, tid = threadIdx.x;
    const uint32_t l = tid & 7u;
    if (p.qstate[q].overflow) {  // (block-uniform)
        if (blockIdx.x == 0 && tid == 0) f.out_counts[q] = 0xFFFFFFFFu;
        return;
    }
    const float* qv = p.qpad + (size_t)q * p.ld;
    const float qmag = p.qinfo[q].qmag;
    const uint32_t count = min(p.qstate[q].cand_count, p.cand_cap);
    constexpr uint32_t kPer = kTailThreads / 8;  // candidates per workgroup step: eight lanes each, in the reference's order
    for (uint32_t c0 = blockIdx.x * kPer; c0 < count; c0 += gridDim.x * kPer) {
        const uint32_t c = c0 + (tid >> 3);
        const uint32_t cc = c < count ? c : count - 1;  // keep every 8-lane group converged
        const uint32_t row = p.cand_rows[(size_t)q * p.cand_cap + cc];
        const float vmag = p.metric == NMN_METRIC_COSINE ? p.norms[row] : 1.0f;
        const float sc = exact_score(qv, p.corpus + (uint64_t)row * p.ld, p.dim, qmag, vmag, p.metric, l);
        if (c < count && l == 0) p.cand_scores[(size_t)q * p.cand_cap + c] = sc;
    }
    __threadfence();  // this thread's scores are visible device-wide before the ticket says so
    __syncthreads();
    if (tid == 0) {
        const uint32_t t = atomicAdd(ticket + q, 1u);
        s_misc[0] = (t == gridDim.x - 1u) ? 1u : 0u;
        if (t == gridDim.x - 1u) ticket[q] = 0u;  // (everybody else has come and gone: ready for the next launch on this stream)
    }
    __syncthreads();
    if (s_misc[0] == 0u) return;
    __threadfence();  // acquire: the other workgroups' scores
    uint32_t n = min(count, (uint32_t)NMN_MAX_TOP_K);
    for (uint32_t i = tid; i < n; i += kTailThreads) {
        const uint32_t row = f.cand_rows[(size_t)q * f.cand_cap + i];
        const uint32_t key = score_to_key(__builtin_nontemporal_load(p.cand_scores + (size_t)q * p.cand_cap + i));
        list[i] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - row);
    }
    __syncthreads();
    constexpr uint32_t kShort = 1024;  // lists up to this long are ordered by runs and ranks (two entries per thread here)
    if (n > kShort && f.k < kShort) {  // (block-uniform) only the k best are wanted: final_kernel's pre-pick
        __shared__ unsigned long long top[kShort];
        const uint32_t T = radix2<kTailThreads>([&](uint32_t e) { return (uint32_t)(list[e] >> 32); }, n, f.k, hist, &pick);
        if (tid == 0) s_misc[1] = 0;
        __syncthreads();
        for (uint32_t b0 = tid & ~63u; b0 < n; b0 += kTailThreads) {
            const uint32_t i = b0 + (tid & 63u);
            const unsigned long long v = i < n ? list[i] : 0ull;
            const bool pr = i < n && (uint32_t)(v >> 32) >= T;
            const uint32_t pos = wave_append(pr, &s_misc[1]);
            if (pr && pos < kShort) top[pos] = v;
        }
        __syncthreads();
        const uint32_t c = s_misc[1];
        if (c <= kShort) {
            for (uint32_t i = tid; i < c; i += kTailThreads) list[i] = top[i];
            n = c;
        }
        __syncthreads();
    }
    sort_and_emit(list, n, n, f.k, f.row_base, f.out_rows + (size_t)q * f.k, f.out_scores + (size_t)q * f.k, f.out_counts + q);
}

hipError_t launch_rescore_final(const RescoreParams& p, const FinalParams& f, uint32_t* ticket, hipStream_t s) {
    // cand_cap candidates = cand_cap / 128 workgroup steps; every workgroup of a query's row takes a ticket, so the row is short
    const uint32_t gx = std::max<uint32_t>(1u, std::min<uint32_t>(32u, (p.cand_cap + kTailThreads / 8 - 1) / (kTailThreads / 8)));
    hipLaunchKernelGGL(rescore_final_kernel, dim3(gx, p.nq), dim3(kTailThreads), 0, s, p, f, ticket);
    return hipGetLastError();
}

// ---- exact score of explicit rows ------------------------------------------------------------
__global__ void __launch_bounds__(256) score_rows_kernel(const float* __restrict__ corpus,
                                                         const float* __restrict__ norms,
                                                         const float* __restrict__ qpad,
                                                         const QInfo* __restrict__ qinfo,
                                                         const uint64_t* __restrict__ rows, uint32_t n_rows,
                                                         uint32_t ld, uint32_t dim, int metric,
                                                         float* __restrict__ out) {
    const uint32_t q = blockIdx.y;
    const uint32_t l = threadIdx.x & 7u;
    const uint32_t i = blockIdx.x * 32u + (threadIdx.x >> 3);
    const uint32_t ii = i < n_rows ? i : n_rows - 1;
    const uint64_t row = rows[ii];
    const float* v = corpus + row * (uint64_t)ld;
    const float vmag = metric == NMN_METRIC_COSINE ? norms[row] : 1.0f;
    const float sc = exact_score(qpad + (size_t)q * ld, v, dim, qinfo[q].qmag, vmag, metric, l);
    if (i < n_rows && l == 0) out[(size_t)q * n_rows + i] = sc;
}

hipError_t launch_score_rows(const float* corpus, const float* norms, const float* qpad, const QInfo* qinfo,
                             const uint64_t* rows, uint32_t n_rows, uint32_t nq, uint32_t ld, uint32_t dim,
                             int metric, float* out, hipStream_t s) {
    if (n_rows == 0 || nq == 0) return hipSuccess;
    dim3 grid((n_rows + 31) / 32, nq);
    hipLaunchKernelGGL(score_rows_kernel, grid, dim3(256), 0, s, corpus, norms, qpad, qinfo, rows, n_rows, ld, dim,
                       metric, out);
    return hipGetLastError();
}

// ---- exact scan of every row (fallback path and the count certificate) ------------------------
// Grid-stride over groups of 32 rows; only queries flagged `overflow` are processed when qstate is
// given, so on the normal path this launch costs one early-exit wave per block.
__global__ void __launch_bounds__(256) exact_scan_kernel(ExactScanParams p) {
    const uint32_t q = blockIdx.y;
    if (p.qstate && p.qstate[q].overflow == 0) return;
    const uint32_t l = threadIdx.x & 7u;
    const float* qv = p.qpad + (size_t)q * p.ld;
    const float qmag = p.qinfo[q].qmag;
    const uint64_t n_pad = (p.n_rows + 63) & ~63ull;
    for (uint64_t base = (uint64_t)blockIdx.x * 32u; base < n_pad; base += (uint64_t)gridDim.x * 32u) {
        const uint64_t row = base + (threadIdx.x >> 3);
        bool valid = row < p.n_rows;
        if (valid && p.mask) valid = ((p.mask[row >> 6] >> (row & 63)) & 1ull) != 0;
        // the 8-lane group shares `row`, hence `valid`: the branch is uniform per group
        uint32_t bits = kScoreSentinelBits;
        if (valid) {
            const float vmag = p.metric == NMN_METRIC_COSINE ? p.norms[row] : 1.0f;
            bits = f2u(exact_score(qv, p.corpus + row * (uint64_t)p.ld, p.dim, qmag, vmag, p.metric, l));
        }
        if (l == 0 && row < n_pad) p.scores[score_at(row, q, p.nql)] = bits;
    }
}

hipError_t launch_exact_scan(const ExactScanParams& p, hipStream_t s) {
    if (p.n_rows == 0) return hipSuccess;
    // large shards of whole-stage rows: a lane per row streaming through LDS (nmn_ingest.hip), same scores bit for bit
    static const bool no_exact_rows = getenv("NMN_NO_EXACT_ROWS") != nullptr;
    if (!no_exact_rows && p.n_rows >= (1u << 16) && exact_rows_supported(p.ld, p.dim, p.metric))
        return launch_exact_rows(p.corpus, p.norms, p.ld, p.n_rows, p.qpad, p.qinfo, p.qstate, 2, p.mask, nullptr, p.scores, p.nql, p.nq,
                                 p.metric, s);
    uint64_t blocks = (p.n_rows + 31) / 32;
    if (blocks > 2048) blocks = 2048;
    dim3 grid((unsigned)blocks, p.nq);
    hipLaunchKernelGGL(exact_scan_kernel, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---- the whole SIMILAR TOP-K of a small shard in ONE launch ----------------------------------------------------------------
// The sizes the reference itself publishes (1k-10k rows of 128 floats, vector_engine/benches/vector_engine_bench.rs:40-61;
// lib.rs:4255-4276) are launch-bound on a GPU: the five launches of the pipeline, one H2D and one D2H cost 57 us at
// 10k x 128 while the arithmetic is a microsecond.  Up to 65 536 rows and 64 MiB of corpus one kernel does everything, and
// nothing else crosses PCIe as a separate operation: the query travels in the KERNEL ARGUMENTS (<= 768 floats), the result
// is written by the kernel straight into pinned host memory mapped into the device.
//   * every workgroup: |q| in reference order, the EXACT reference-order score of each of its rows (8 lanes per row as in
//     exact_scan_kernel: at this size the exact pass is as cheap as an approximate one, so there is no margin machinery
//     at all), its rows sorted in LDS, its best min(k, rows) composites to a pool;
//   * the LAST workgroup to finish (ticket counter; release / acquire fences at agent scope) selects the top-k of the pool
//     (in LDS when it fits, else the 64-bit radix select of the exact fallback) and emits it.
struct TinyParams {
    const float* corpus;
    const float* norms;
    const uint64_t* mask;  // nullable, DEVICE
    unsigned long long* pool;  // [grid][kcap] composites (0 = no entry)
    uint32_t* ticket;      // zero between launches (the last workgroup resets it)
    uint64_t* out_rows;    // pinned host memory, mapped
    float* out_scores;
    uint32_t* out_count;
    uint32_t* out_seq;     // written LAST (system-scope release): the host polls it instead of synchronising the stream
    uint64_t n_rows, row_base;
    uint32_t ld, dim, k, kcap, rows_per_wg, seq;
    int metric;
    float q[kTinyMaxDim];
};

__global__ void __launch_bounds__(kSelThreads) tiny_search_kernel(TinyParams p) {
    __shared__ unsigned long long list[NMN_MAX_TOP_K];
    __shared__ uint32_t hist[kBins];
    __shared__ PickResult pick;
    __shared__ uint32_t s_misc[2];
    __shared__ float qlds[kTinyMaxDim];
    __shared__ float s_qmag;
    __shared__ uint32_t s_ticket;
    const uint32_t tid = threadIdx.x, l = tid & 7u, wg = blockIdx.x, G = gridDim.x;
    for (uint32_t i = tid; i < kTinyMaxDim; i += kSelThreads) qlds[i] = i < p.dim ? p.q[i] : 0.0f;
    __syncthreads();
    if (tid < 8) {
        const float ss = dot8_group<32>(qlds, qlds, p.dim, l);
        if (tid == 0) s_qmag = sqrt_rn(ss);
    }
    __syncthreads();
    const float qmag = s_qmag;
    // exact score of this workgroup's rows -> composites in LDS (0 = row does not take part)
    const uint64_t r0 = (uint64_t)wg * p.rows_per_wg;
    const uint32_t R = (uint32_t)min((uint64_t)p.rows_per_wg, p.n_rows > r0 ? p.n_rows - r0 : 0ull);
    const bool small_k = p.k <= 64u;  // (kernel-argument uniform; kcap <= k) the wave-level top-64 network serves this launch
    uint32_t np2 = 1;
    while (np2 < max(R, 1u)) np2 <<= 1;
    if (small_k) np2 = max((R + 63u) & ~63u, 64u);  // here: the slots scored (whole waves' worth; not a power of two)
    for (uint32_t base = 0; base < np2; base += kSelThreads / 8) {
        const uint32_t ri = base + (tid >> 3);
        const uint64_t row = r0 + ri;
        bool valid = ri < R;
        if (valid && p.mask) valid = ((p.mask[row >> 6] >> (row & 63)) & 1ull) != 0;
        unsigned long long c = 0ull;
        if (valid) {  // (uniform per 8-lane group)
            const float vmag = p.metric == NMN_METRIC_COSINE ? p.norms[row] : 1.0f;
            const float sc = exact_score(qlds, p.corpus + row * (uint64_t)p.ld, p.dim, qmag, vmag, p.metric, l);
            c = ((unsigned long long)score_to_key(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)row);
        }
        if (l == 0 && ri < np2) list[ri] = c;
    }
    __syncthreads();
    // this workgroup's rows in final order (score desc, row asc); its first kcap go to the pool
    if (small_k) {
        const unsigned long long v = wg_top64(list, np2 >> 6);
        if (tid < p.kcap) p.pool[(size_t)wg * p.kcap + tid] = v;
    } else {
    for (uint32_t size = 2; size <= np2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < (np2 >> 1); t += kSelThreads) {
                const uint32_t lo = ((t / stride) * stride * 2u) + (t % stride), hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = list[lo], b = list[hi];
                if ((a < b) == desc) {
                    list[lo] = b;
                    list[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < p.kcap; i += kSelThreads) p.pool[(size_t)wg * p.kcap + i] = i < np2 ? list[i] : 0ull;
    }
    // last workgroup standing merges (release: pool stores visible device-wide before the ticket; acquire before reading)
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        s_ticket = atomicAdd(p.ticket, 1u);
    }
    __syncthreads();
    if (s_ticket != G - 1) return;
    if (tid == 0) {
        __threadfence();
        *p.ticket = 0u;  // for the next launch on this workspace (stream order)
    }
    __syncthreads();
    const uint32_t total = G * p.kcap;
    if (small_k && total <= kSelThreads) {
        const uint32_t n64 = (total + 63u) >> 6;
        if (tid < n64 * 64u) list[tid] = tid < total ? __hip_atomic_load(&p.pool[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        __syncthreads();
        const unsigned long long v = wg_top64(list, n64);
        emit_top64(v, p.k, p.row_base, p.out_rows, p.out_scores, p.out_count);
    } else if (total <= NMN_MAX_TOP_K) {
        for (uint32_t i = tid; i < total; i += kSelThreads)
            list[i] = __hip_atomic_load(&p.pool[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) s_misc[0] = 0;
        __syncthreads();
        uint32_t live = 0;
        for (uint32_t i = tid; i < total; i += kSelThreads) live += list[i] != 0ull;
        if (live) atomicAdd(&s_misc[0], live);
        __syncthreads();
        sort_and_emit(list, total, s_misc[0], p.k, p.row_base, p.out_rows, p.out_scores, p.out_count);  // (empty slots sort last)
    } else {
        const uint32_t n_round = (total + kSelThreads - 1) / kSelThreads * kSelThreads;
        const uint32_t n = exact_select_walk(
            [&](auto&& f) {
                for (uint32_t e = tid; e < n_round; e += kSelThreads) {
                    const unsigned long long c = e < total ? __hip_atomic_load(&p.pool[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                    f((uint64_t)(0xFFFFFFFFu - (uint32_t)c), c ? (uint32_t)(c >> 32) : kKeyMasked);
                }
            },
            p.k, list, hist, &pick, s_misc);
        __syncthreads();
        sort_and_emit(list, n, n, p.k, p.row_base, p.out_rows, p.out_scores, p.out_count);
    }
    // publish: every result store of this workgroup is ordered before the sequence word the host is spinning on
    __syncthreads();
    if (tid == 0) {
        __threadfence_system();
        __hip_atomic_store(p.out_seq, p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

bool tiny_supported(uint64_t n_rows, uint32_t ld, uint32_t dim, uint32_t k) {
    return n_rows >= 1 && n_rows <= 65536 && dim <= (uint32_t)kTinyMaxDim && k <= 1024u && n_rows * (uint64_t)ld * 4ull <= (64ull << 20);
}
// rows per workgroup / grid for a shard of n_rows (the pool holds grid * kcap composites, kcap = min(k, rows per workgroup))
void tiny_geometry(uint64_t n_rows, uint32_t k, uint32_t* grid, uint32_t* rows_per_wg, uint32_t* kcap) {
    // 128 rows per workgroup = one scoring pass of its 1024 threads (8 lanes per row), up to 128 workgroups; beyond 16k rows
    // the workgroups take more passes
    uint32_t g = (uint32_t)std::min<uint64_t>((n_rows + 127) / 128, 128);
    g = std::max(g, 1u);
    const uint32_t per = (uint32_t)((n_rows + g - 1) / g);
    *grid = (uint32_t)((n_rows + per - 1) / per);
    *rows_per_wg = per;
    *kcap = std::min(k, per);
}

hipError_t launch_tiny_search(const float* corpus, const float* norms, const uint64_t* mask_dev, uint64_t n_rows, uint64_t row_base,
                              uint32_t ld, uint32_t dim, uint32_t k, int metric, const float* query_host, unsigned long long* pool,
                              uint32_t* ticket, uint64_t* out_rows, float* out_scores, uint32_t* out_count, uint32_t* out_seq, uint32_t seq,
                              hipStream_t s) {
    TinyParams p{};
    p.out_seq = out_seq;
    p.seq = seq;
    p.corpus = corpus;
    p.norms = norms;
    p.mask = mask_dev;
    p.pool = pool;
    p.ticket = ticket;
    p.out_rows = out_rows;
    p.out_scores = out_scores;
    p.out_count = out_count;
    p.n_rows = n_rows;
    p.row_base = row_base;
    p.ld = ld;
    p.dim = dim;
    p.k = k;
    p.metric = metric;
    uint32_t grid = 1;
    tiny_geometry(n_rows, k, &grid, &p.rows_per_wg, &p.kcap);
    memcpy(p.q, query_host, (size_t)dim * sizeof(float));
    hipLaunchKernelGGL(tiny_search_kernel, dim3(grid), dim3(kSelThreads), 0, s, p);
    return hipGetLastError();
}

}  // namespace nmn
