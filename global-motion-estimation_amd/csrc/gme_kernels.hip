// Global-motion-estimation kernels for gfx950: pyramid, first parameters, per-level
// robust fit (model field, outlier mask, normal-equation sums), compensation, SSE.
//
// Replaces motion.py:109-341 and utils.py:34-51,100-116 of the reference, except the
// two 3x3 solves per level (motion.py:262-264,280-282), which stay on the host.
//
// Floating point: every float64 operation below is written with the *_rn intrinsics,
// so no FMA contraction can change a rounding; the sums reproduce the reference's
// sequential `acc += (product) * w` order bit for bit.
#include <stdlib.h>

#include "gme_internal.h"

namespace {

// ---------------------------------------------------------------------------
// cv2.pyrDown (utils.py:48): 5x5 [1 4 6 4 1]^2, BORDER_REFLECT_101, (s + 128) >> 8
// ---------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

// Each thread produces 4 horizontally adjacent output pixels of two rows.  k_pyrdown covers the interior quads
// (x in [4, x_end)): per source row the 16 aligned bytes [2x-4, 2x+12) hold the 11 taps
// 2x-2 .. 2x+8 (one dword-aligned 16-byte load instead of 20 byte loads).  The quads that touch the
// left/right border need BORDER_REFLECT_101 per tap; they run in their own small launch
// (k_pyrdown_edge) so that no interior wave has to execute the slow path as well.
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ int pyr_interior_end(int sW, int dW)
{
    // largest multiple of 4, x_end, such that every quad x in [4, x_end) has 2x+12 <= sW and x+4 <= dW
    int e = min((sW - 12) / 2 + 4, dW) & ~3;
    return e < 4 ? 4 : e;
}

__global__ void __launch_bounds__(256) k_pyrdown(const uint8_t* src, long long src_stride, int sH, int sW,
                                                  int spitch, uint8_t* dst, long long dst_stride, int dH,
                                                  int dW, int dpitch)
{
    // a thread produces the quad x .. x+3 of output rows y and y + 1: their 5-row supports overlap in 3 source rows, so
    // 7 row loads and 7 horizontal passes serve both (10 with one output row per thread)
    const int x = 4 + (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 2;
    if (x >= pyr_interior_end(sW, dW) || y >= dH) return;
    const uint8_t* s = src + (long long)blockIdx.z * src_stride;
    uint8_t* o = dst + (long long)blockIdx.z * dst_stride + (long long)y * dpitch + x;
    const bool two = y + 1 < dH;                        // an odd dH: the last thread row stores one row (its loads stay valid: reflect101)
    // source rows first (one reflection per side is all rows -2 .. sH + 3 of a plane of 4 or more rows need; no loop, no
    // branch between the seven loads), then the loads, then the arithmetic
    int sr[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        int q = 2 * y + r - 2;
        q = q < 0 ? -q : q;
        q = q >= sH ? 2 * (sH - 1) - q : q;
        sr[r] = sH >= 4 ? q : reflect101(2 * y + r - 2, sH);
    }
    u32x4_a4 vs[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) vs[r] = *(const u32x4_a4*)(s + (long long)sr[r] * spitch + 2 * x - 4);
    int h[7][4];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const u32x4_a4 v = vs[r];
        // output p takes bytes 2+2p .. 6+2p of the 16: the first four through v_dot4 against the taps (1 4 6 4),
        // the fifth (weight 1) by one bit-field extract
        const uint32_t w4[4] = { __builtin_amdgcn_alignbyte(v.y, v.x, 2u), v.y, __builtin_amdgcn_alignbyte(v.z, v.y, 2u), v.z };
        const uint32_t last[4] = { (v.y >> 16) & 0xFFu, v.z & 0xFFu, (v.z >> 16) & 0xFFu, v.w & 0xFFu };
#pragma unroll
        for (int p = 0; p < 4; ++p) h[r][p] = (int)__builtin_amdgcn_udot4(w4[p], 0x04060401u, last[p], false);
    }
    uint32_t out0 = 0, out1 = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int a0 = h[0][p] + 4 * h[1][p] + 6 * h[2][p] + 4 * h[3][p] + h[4][p];
        const int a1 = h[2][p] + 4 * h[3][p] + 6 * h[4][p] + 4 * h[5][p] + h[6][p];
        out0 |= (uint32_t)((a0 + 128) >> 8) << (8 * p);
        out1 |= (uint32_t)((a1 + 128) >> 8) << (8 * p);
    }
    *(uint32_t*)o = out0;
    if (two) *(uint32_t*)(o + dpitch) = out1;
}

// One launch per level for planes whose width is a multiple of 8 (every level of the BASELINE shapes): a workgroup
// stages the 2 T + 3 source rows of T output rows in LDS -- rows reflected (BORDER_REFLECT_101) when it picks them,
// the two border columns on each side written into an apron behind the barrier -- so every output quad, border ones
// included, runs the interior arithmetic of k_pyrdown on LDS reads.  No border launch (k_pyrdown_edge16 was a quarter
// of the pyramid's time for 2 % of its pixels), each source byte crosses HBM once (+ 3 halo rows per 16).
constexpr int PYR_T = 8;                       // output rows per workgroup
constexpr int PYR_ROWS = 2 * PYR_T + 3;        // source rows staged
constexpr int PYR_APRON = 4;                   // pixel 0 sits at byte 4 of an LDS row: quad x reads bytes 2x .. 2x+15 (8-byte aligned)

__global__ void __launch_bounds__(256) k_pyrdown_lds(const uint8_t* src, long long src_stride, int sH, int sW, int spitch,
                                                      uint8_t* dst, long long dst_stride, int dH, int dW, int dpitch, int lpitch,
                                                      uint32_t magic_row, uint32_t magic_quads)
{
    extern __shared__ uint32_t pyr_lds[];
    uint8_t* rows = (uint8_t*)pyr_lds;         // [PYR_ROWS][lpitch], lpitch % 8 == 0
    const int y0 = blockIdx.y * PYR_T;
    const uint8_t* s = src + (long long)blockIdx.z * src_stride;
    // ---- stage: 16-byte segments, source row r of the tile = reflect101(2 y0 - 2 + r); four loads in flight per thread,
    // index -> (row, segment) by a multiply-shift (magic_row = 2^20 / per_row + 1: exact below 4096 items)
    const int segs = sW >> 4, tail = sW & 15;   // sW % 8 == 0: an 8-byte tail segment when sW % 16 == 8
    const int per_row = segs + (tail ? 1 : 0);
    const int total = PYR_ROWS * per_row;
    for (int base = threadIdx.x; base < total; base += 4 * 256) {
        u32x4_a4 v[4];
        int off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int it = base + 256 * u;
            const int r = (int)(__umul24((uint32_t)it, magic_row) >> 20), sg = it - r * per_row;
            off[u] = -1;
            v[u] = u32x4_a4{ 0, 0, 0, 0 };
            if (it < total) {
                const uint8_t* g = s + (long long)reflect101(2 * y0 - 2 + r, sH) * spitch + 16 * sg;
                off[u] = r * lpitch + PYR_APRON + 16 * sg + (sg < segs ? 0 : 1);      // odd: the 8-byte tail segment
                if (sg < segs) v[u] = *(const u32x4_a4*)g;
                else { const uint2 t = *(const uint2*)g; v[u].x = t.x; v[u].y = t.y; }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (off[u] >= 0) {
                uint32_t* l = (uint32_t*)(rows + (off[u] & ~1));
                l[0] = v[u].x; l[1] = v[u].y;
                if (!(off[u] & 1)) { l[2] = v[u].z; l[3] = v[u].w; }
            }
    }
    __syncthreads();
    // ---- aprons: columns -2, -1 = 2, 1 and sW, sW + 1 = sW - 2, sW - 3 (reflect101; sW >= 8)
    if (threadIdx.x < PYR_ROWS * 4) {
        const int r = threadIdx.x >> 2, k = threadIdx.x & 3;
        uint8_t* row = rows + r * lpitch + PYR_APRON;
        const int to = k == 0 ? -2 : k == 1 ? -1 : k == 2 ? sW : sW + 1;
        const int from = k == 0 ? 2 : k == 1 ? 1 : k == 2 ? sW - 2 : sW - 3;
        row[to] = row[from];
    }
    __syncthreads();
    // ---- compute: item = (row pair, quad); the arithmetic of k_pyrdown
    const int quads = dW >> 2;                  // dW = sW / 2 is a multiple of 4
    for (int it = threadIdx.x; it < (PYR_T / 2) * quads; it += 256) {
        const int rp = (int)(__umul24((uint32_t)it, magic_quads) >> 20), qx = it - rp * quads;
        const int y = y0 + 2 * rp;
        if (y >= dH) break;                     // items are row-major: everything behind is outside too
        const uint8_t* l = rows + (4 * rp) * lpitch + 8 * qx;        // byte 2x - 4 + PYR_APRON of tile row 4 rp, x = 4 qx
        int h[7][4];
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const uint2 lo = *(const uint2*)(l + r * lpitch), hi = *(const uint2*)(l + r * lpitch + 8);
            const uint32_t w4[4] = { __builtin_amdgcn_alignbyte(lo.y, lo.x, 2u), lo.y, __builtin_amdgcn_alignbyte(hi.x, lo.y, 2u), hi.x };
            const uint32_t last[4] = { (lo.y >> 16) & 0xFFu, hi.x & 0xFFu, (hi.x >> 16) & 0xFFu, hi.y & 0xFFu };
#pragma unroll
            for (int p = 0; p < 4; ++p) h[r][p] = (int)__builtin_amdgcn_udot4(w4[p], 0x04060401u, last[p], false);
        }
        uint32_t out0 = 0, out1 = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int a0 = h[0][p] + 4 * h[1][p] + 6 * h[2][p] + 4 * h[3][p] + h[4][p];
            const int a1 = h[2][p] + 4 * h[3][p] + 6 * h[4][p] + 4 * h[5][p] + h[6][p];
            out0 |= (uint32_t)((a0 + 128) >> 8) << (8 * p);
            out1 |= (uint32_t)((a1 + 128) >> 8) << (8 * p);
        }
        uint8_t* o = dst + (long long)blockIdx.z * dst_stride + (long long)y * dpitch + 4 * qx;
        *(uint32_t*)o = out0;
        if (y + 1 < dH) *(uint32_t*)(o + dpitch) = out1;
    }
}

// border pixels: x in [0, 4) and [x_end, dW); one thread per pixel
__global__ void __launch_bounds__(256) k_pyrdown_edge(const uint8_t* src, long long src_stride, int sH, int sW,
                                                       int spitch, uint8_t* dst, long long dst_stride, int dH,
                                                       int dW, int dpitch)
{
    const int x_end = pyr_interior_end(sW, dW);
    const int n_left = min(4, dW), n_right = max(0, dW - x_end);
    const int per_row = n_left + n_right;
    const int id = blockIdx.x * 256 + threadIdx.x;
    const int y = id / per_row, e = id - y * per_row;      // a wave covers 64 / per_row rows: few cache lines per load
    if (y >= dH) return;
    const int x = e < n_left ? e : x_end + (e - n_left);
    const uint8_t* s = src + (long long)blockIdx.z * src_stride;
    const int k[5] = { 1, 4, 6, 4, 1 };
    int cx[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) cx[d] = reflect101(2 * x + d - 2, sW);
    int a = 0;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
        const uint8_t* row = s + (long long)reflect101(2 * y + dy - 2, sH) * spitch;
        int h = 0;
#pragma unroll
        for (int d = 0; d < 5; ++d) h += k[d] * row[cx[d]];
        a += k[dy] * h;
    }
    dst[(long long)blockIdx.z * dst_stride + (long long)y * dpitch + x] = (uint8_t)((a + 128) >> 8);
}

// Border pixels for rows of >= 16 bytes whose width is a multiple of 4 (every pyramid level of the two BASELINE
// shapes): all taps of a left-border pixel lie in the row's first 16 bytes, all taps of a right-border pixel in its
// last 16 (2 x_end - 2 >= sW - 16, reflections included) -- five 16-byte loads per thread instead of 25 byte loads
// that each touch their own cache line; the tap is picked out of the four dwords with a run-time byte index.
__device__ __forceinline__ int byte_of(const u32x4_a4& v, int idx)
{
    const uint32_t lo = idx < 8 ? (idx < 4 ? v.x : v.y) : (idx < 12 ? v.z : v.w);
    return (int)__builtin_amdgcn_ubfe(lo, (uint32_t)(idx & 3) * 8u, 8u);
}

__global__ void __launch_bounds__(256) k_pyrdown_edge16(const uint8_t* src, long long src_stride, int sH, int sW,
                                                         int spitch, uint8_t* dst, long long dst_stride, int dH,
                                                         int dW, int dpitch)
{
    const int x_end = pyr_interior_end(sW, dW);
    const int n_left = min(4, dW), n_right = max(0, dW - x_end);
    const int per_row = n_left + n_right;
    const int id = blockIdx.x * 256 + threadIdx.x;
    const int y = id / per_row, e = id - y * per_row;
    if (y >= dH) return;
    const bool left = e < n_left;
    const int x = left ? e : x_end + (e - n_left);
    const int base = left ? 0 : sW - 16;
    const uint8_t* s = src + (long long)blockIdx.z * src_stride + base;
    const int k[5] = { 1, 4, 6, 4, 1 };
    int cx[5];
#pragma unroll
    for (int d = 0; d < 5; ++d) cx[d] = reflect101(2 * x + d - 2, sW) - base;      // 0 .. 15
    int a = 0;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
        const u32x4_a4 v = *(const u32x4_a4*)(s + (long long)reflect101(2 * y + dy - 2, sH) * spitch);
        int h = 0;
#pragma unroll
        for (int d = 0; d < 5; ++d) h += k[d] * byte_of(v, cx[d]);
        a += k[dy] * h;
    }
    dst[(long long)blockIdx.z * dst_stride + (long long)y * dpitch + x] = (uint8_t)((a + 128) >> 8);
}

// ---------------------------------------------------------------------------
// motion.compute_first_parameters (motion.py:176-188)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_first_params(const int32_t* dense, int n, float* params0)
{
    __shared__ long long part[2][4];
    const int32_t* mf = dense + (long long)blockIdx.x * n * 2;
    long long s0 = 0, s1 = 0;
    for (int k = threadIdx.x; k < n; k += 256) { s0 += mf[2 * k]; s1 += mf[2 * k + 1]; }
    for (int m = 32; m > 0; m >>= 1) {
        s0 += ((long long)__shfl_xor((int)(s0 >> 32), m, 64) << 32) | (unsigned)__shfl_xor((int)s0, m, 64);
        s1 += ((long long)__shfl_xor((int)(s1 >> 32), m, 64) << 32) | (unsigned)__shfl_xor((int)s1, m, 64);
    }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = s0; part[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s0 = part[0][0] + part[0][1] + part[0][2] + part[0][3];
        s1 = part[1][0] + part[1][1] + part[1][2] + part[1][3];
        float* o = params0 + (long long)blockIdx.x * 6;
        o[0] = (float)__ddiv_rn((double)s0, (double)n);      // np.mean -> float64, then float32
        o[1] = 0.f; o[2] = 0.f;
        o[3] = (float)__ddiv_rn((double)s1, (double)n);
        o[4] = 0.f; o[5] = 0.f;
    }
}

// motion.parameter_projection (motion.py:191-207) of the float32 first parameters on the device: p[0] and p[3] doubled in
// float32 (exact), the vector widened to the float64 the level-1 fit reads -- what the host does between
// gme_seq_gme_begin and gme_seq_gme_fit(1), without the trip (gme_seq_gme_begin_fit)
__global__ void __launch_bounds__(256) k_project_first(const float* params0, int n, double* params_in)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int k = i % 6;
    const float v = params0[i];
    params_in[i] = (double)((k == 0 || k == 3) ? v * 2.0f : v);
}

// ---------------------------------------------------------------------------
// Per-pair summary row of a motion field -- what a sharded block-matching run exchanges instead of the
// fields themselves (gme_seq_mv_summary[_gather]): float64[6] =
//   modal vector (x, y), the number of blocks that chose it, sum of x, sum of y, checksum.
// The mode is taken over the vectors inside [-64, 64)^2 (both BASELINE windows: bbme.py:146-149 spans
// [-sw, sw + bs)); ties go to the smaller (x + 64) * 128 + (y + 64).  The checksum is
// sum_i ((i mod 251) + 1) * (3 x_i + 5 y_i) over the blocks in row-major order: integer, order-free.
// ---------------------------------------------------------------------------
constexpr int SUMMARY_HALF = 64, SUMMARY_SIDE = 2 * SUMMARY_HALF, SUMMARY_BINS = SUMMARY_SIDE * SUMMARY_SIDE;

__global__ void __launch_bounds__(256) k_mv_summary(const int32_t* mf_all, int n, double* rows)
{
    extern __shared__ unsigned int hist[];                 // SUMMARY_BINS counters
    __shared__ long long acc[3];
    __shared__ unsigned long long best;
    const int32_t* mf = mf_all + (long long)blockIdx.x * n * 2;
    for (int b = threadIdx.x; b < SUMMARY_BINS; b += 256) hist[b] = 0;
    if (threadIdx.x < 3) acc[threadIdx.x] = 0;
    if (threadIdx.x == 0) best = 0;
    __syncthreads();
    long long sx = 0, sy = 0, ck = 0;
    for (int k = threadIdx.x; k < n; k += 256) {
        const int x = mf[2 * k], y = mf[2 * k + 1];
        sx += x; sy += y;
        ck += (long long)(k % 251 + 1) * (3 * (long long)x + 5 * (long long)y);
        if (x >= -SUMMARY_HALF && x < SUMMARY_HALF && y >= -SUMMARY_HALF && y < SUMMARY_HALF)
            atomicAdd(&hist[(x + SUMMARY_HALF) * SUMMARY_SIDE + (y + SUMMARY_HALF)], 1u);
    }
    atomicAdd((unsigned long long*)&acc[0], (unsigned long long)sx);
    atomicAdd((unsigned long long*)&acc[1], (unsigned long long)sy);
    atomicAdd((unsigned long long*)&acc[2], (unsigned long long)ck);
    __syncthreads();
    unsigned long long key = 0;                            // count << 16 | (0xFFFF - bin): larger count, then smaller bin
    for (int b = threadIdx.x; b < SUMMARY_BINS; b += 256) {
        const unsigned long long kb = ((unsigned long long)hist[b] << 16) | (unsigned)(0xFFFF - b);
        if (hist[b] && kb > key) key = kb;
    }
    if (key) atomicMax(&best, key);
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = rows + (long long)blockIdx.x * 6;
        const unsigned long long kb = best;
        const int bin = 0xFFFF - (int)(kb & 0xFFFF);
        o[0] = kb ? (double)(bin / SUMMARY_SIDE - SUMMARY_HALF) : 0.0;
        o[1] = kb ? (double)(bin % SUMMARY_SIDE - SUMMARY_HALF) : 0.0;
        o[2] = (double)(kb >> 16);
        o[3] = (double)acc[0];
        o[4] = (double)acc[1];
        o[5] = (double)acc[2];
    }
}

// ---------------------------------------------------------------------------
// motion.affine_model / get_motion_field_affine (motion.py:91-105,139-157)
// ---------------------------------------------------------------------------
__device__ __forceinline__ int16_t model_component(double p0, double p1, double p2, int i, int j)
{
    // gemv order of the reference's NumPy/OpenBLAS (see oracle/gme_oracle.py affine_field):
    // (p0 + p2*j) + p1*i, each product and sum rounded separately.
    const double d = __dadd_rn(__dadd_rn(p0, __dmul_rn(p2, (double)j)), __dmul_rn(p1, (double)i));
    return (int16_t)(long long)rint(d);      // round-half-even, int16 store wraps
}

// ---------------------------------------------------------------------------
// Opt-in (GME_DEVICE_SOLVE=1, gme_seq_gme_device_solve): the two 3x3 solves of motion.py:262-264,280-282 on the device, so
// that a whole estimate is ONE host round trip instead of three.  The reference inverts F with LAPACK (np.linalg.inv); a
// device solve cannot reproduce LAPACK's last bits (builds already differ by ~1e-13 among themselves), so:
//   * parameters are promised to rtol 1e-10 only (the tolerance every parity test uses for them anyway);
//   * everything DOWNSTREAM of a solve must stay bit-equal -- model fields, masks, compensated frames.  Those depend on
//     the parameters only through round-half-even of the model displacements (model_component below), so the kernel
//     evaluates the field the parameters will be used for and raises the pair's flag when any displacement lies within
//     `margin` (1e-9) of k + 0.5: such a pair -- and a singular system, for which upstream raises LinAlgError -- is
//     redone through the host path by the caller.  Elsewhere a 1e-10 relative parameter error cannot move a rounding.
// One workgroup per pair: thread 0 solves (Gaussian elimination with partial pivoting, every product and sum rounded
// separately), all threads scan the field.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool solve3(const double* F, const double* S, double* x)
{
    double a[3][4];
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) a[r][c] = F[3 * r + c]; a[r][3] = S[r]; }
    for (int k = 0; k < 3; ++k) {
        int piv = k;
        for (int r = k + 1; r < 3; ++r) if (fabs(a[r][k]) > fabs(a[piv][k])) piv = r;
        if (a[piv][k] == 0.0) return false;
        if (piv != k) for (int c = 0; c < 4; ++c) { const double t = a[k][c]; a[k][c] = a[piv][c]; a[piv][c] = t; }
        for (int r = k + 1; r < 3; ++r) {
            const double m = __ddiv_rn(a[r][k], a[k][k]);
            for (int c = k; c < 4; ++c) a[r][c] = __dsub_rn(a[r][c], __dmul_rn(m, a[k][c]));
        }
    }
    for (int k = 2; k >= 0; --k) {
        double t = a[k][3];
        for (int c = k + 1; c < 3; ++c) t = __dsub_rn(t, __dmul_rn(a[k][c], x[c]));
        x[k] = __ddiv_rn(t, a[k][k]);
    }
    return true;
}

__global__ void __launch_bounds__(256) k_solve3(const double* sums, int pairs, int project, int h, int w, double margin,
                                                double* params_out, int32_t* flags, int flag_bit)
{
    __shared__ double p[6];
    __shared__ int bad;
    const int pair = blockIdx.x;
    if (threadIdx.x == 0) {
        const double* s = sums + (long long)pair * 15;
        double ax[3] = { 0, 0, 0 }, ay[3] = { 0, 0, 0 };
        bad = (solve3(s, s + 9, ax) && solve3(s, s + 12, ay)) ? 0 : 4;           // 4: singular system (numpy.linalg.LinAlgError upstream)
        if (project) { ax[0] = __dmul_rn(ax[0], 2.0); ay[0] = __dmul_rn(ay[0], 2.0); }     // motion.py:191-207
        p[0] = ax[0]; p[1] = ax[1]; p[2] = ax[2]; p[3] = ay[0]; p[4] = ay[1]; p[5] = ay[2];
        for (int k = 0; k < 6; ++k) params_out[(long long)pair * 6 + k] = p[k];
    }
    __syncthreads();
    int near_half = 0;
    for (int b = threadIdx.x; b < h * w; b += blockDim.x) {
        const int i = b / w, j = b - i * w;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const double d = __dadd_rn(__dadd_rn(p[3 * c], __dmul_rn(p[3 * c + 2], (double)j)), __dmul_rn(p[3 * c + 1], (double)i));
            const double fr = d - floor(d);
            if (fabs(fr - 0.5) < margin || !(fabs(d) < 30000.0)) near_half = 1;    // (also: outside int16's comfortable range, NaN)
        }
    }
    if (near_half) atomicOr(&bad, flag_bit);
    __syncthreads();
    if (threadIdx.x == 0 && bad) atomicOr(flags + pair, bad);
}


__global__ void __launch_bounds__(256) k_affine_field(const double* params, int h, int w, int16_t* out)
{
    const int n = h * w;
    const double* p = params + (long long)blockIdx.y * 6;
    int16_t* o = out + (long long)blockIdx.y * n * 2;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int i = k / w, j = k - i * w;
    o[2 * k] = model_component(p[0], p[1], p[2], i, j);
    o[2 * k + 1] = model_component(p[3], p[4], p[5], i, j);
}

// ---------------------------------------------------------------------------
// motion.best_affine_parameters_robust minus BBME and solve (motion.py:232-279)
// one workgroup per pair
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_fit_level(const int32_t* gt_all, int h, int w, const double* params,
                                                    int drop, double wgt, int16_t* model_all,
                                                    uint8_t* mask_all, int32_t* diff_all, int32_t* thr_all,
                                                    double* sums_all, int4* list_all, int list_lds)
{
    extern __shared__ int4 dyn_lds[];
    __shared__ unsigned hist[256];
    __shared__ unsigned sel_prefix, sel_rank;
    const int n = h * w;
    const int pair = blockIdx.x;
    const int32_t* gt = gt_all + (long long)pair * n * 2;
    int16_t* model = model_all + (long long)pair * n * 2;
    uint8_t* mask = mask_all + (long long)pair * n;
    int32_t* diff = diff_all + (long long)pair * n;
    const double* p = params + (long long)pair * 6;
    const double p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4], p5 = p[5];

    // model field and L1 difference (motion.py:232-239)
    for (int k = threadIdx.x; k < n; k += 256) {
        const int i = k / w, j = k - i * w;
        const int16_t m0 = model_component(p0, p1, p2, i, j), m1 = model_component(p3, p4, p5, i, j);
        model[2 * k] = m0; model[2 * k + 1] = m1;
        diff[k] = abs(gt[2 * k] - (int)m0) + abs(gt[2 * k + 1] - (int)m1);
    }
    // threshold = sorted(diff)[n - drop], or sorted(diff)[0] when drop == 0 (motion.py:240-243);
    // drop < 0 asks for the unmasked fit of motion.best_affine_parameters (motion.py:33-88)
    if (threadIdx.x == 0) { sel_prefix = drop < 0 ? 0x7FFFFFFFu : 0u; sel_rank = drop <= 0 ? 0 : (unsigned)(n - drop); }
    __syncthreads();
    for (int shift = drop < 0 ? -1 : 24; shift >= 0; shift -= 8) {
        hist[threadIdx.x] = 0;
        __syncthreads();
        const unsigned prefix = sel_prefix;
        const unsigned himask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int k = threadIdx.x; k < n; k += 256) {
            const unsigned v = (unsigned)diff[k];
            if ((v & himask) == prefix) atomicAdd(&hist[(v >> shift) & 255], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned rank = sel_rank, cum = 0;
            int b = 0;
            for (; b < 255; ++b) {
                if (rank < cum + hist[b]) break;
                cum += hist[b];
            }
            sel_rank = rank - cum;
            sel_prefix = prefix | ((unsigned)b << shift);
        }
        __syncthreads();
    }
    const int thr = (int)sel_prefix;
    if (threadIdx.x == 0) thr_all[pair] = thr;

    // mask (strict >, motion.py:244) and ordered compaction of the inliers: entry e of the
    // list is the e-th inlier in row-major order, stored as (4i, 4j, gt0, gt1)
    __shared__ int wave_count[4];
    __shared__ int list_base;
    int4* list = list_lds ? (int4*)dyn_lds : list_all + (long long)pair * n;
    if (threadIdx.x == 0) list_base = 0;
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += 256) {
        const int k = k0 + threadIdx.x;
        bool inl = false;
        if (k < n) {
            const bool out = diff[k] > thr;
            mask[k] = out;
            inl = !out;
        }
        const unsigned long long bal = __ballot(inl);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) wave_count[wv] = __popcll(bal);
        __syncthreads();
        int off = list_base;
        for (int q = 0; q < wv; ++q) off += wave_count[q];
        if (inl) {
            const int i = k / w, j = k - i * w;
            list[off + __popcll(bal & ((1ull << lane) - 1))] = make_int4(i * 4, j * 4, gt[2 * k], gt[2 * k + 1]);
        }
        __syncthreads();
        if (threadIdx.x == 0) list_base += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        __syncthreads();
    }
    const int m = list_base;

    // sequential float64 sums over inliers in row-major order (motion.py:248-261,266-279):
    // twelve independent chains F00 F01 F02 F11 F12 F22 | Sx0..2 | Sy0..2.  Only the additions
    // are order sensitive, so all 256 threads form the terms (product * w, rounded like the
    // reference) of 256 list entries at a time into LDS, then twelve lanes add them in list order.
    __shared__ double terms[12][256];
    const int fa[6] = { 0, 0, 0, 1, 1, 2 }, fb[6] = { 0, 1, 2, 1, 2, 2 };
    double acc = 0.0;
    for (int e0 = 0; e0 < m; e0 += 256) {
        const int e = e0 + threadIdx.x;
        if (e < m) {
            const int4 v = list[e];
            const double one = 1.0, x = (double)v.x, y = (double)v.y, g0 = (double)v.z, g1 = (double)v.w;
            const double va[3] = { one, x, y };
#pragma unroll
            for (int c = 0; c < 6; ++c) terms[c][threadIdx.x] = __dmul_rn(__dmul_rn(va[fa[c]], va[fb[c]]), wgt);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                terms[6 + a][threadIdx.x] = __dmul_rn(__dmul_rn(va[a], g0), wgt);
                terms[9 + a][threadIdx.x] = __dmul_rn(__dmul_rn(va[a], g1), wgt);
            }
        }
        __syncthreads();
        if (threadIdx.x < 12) {
            const int cnt = min(256, m - e0);
            const double* t = terms[threadIdx.x];
            for (int u = 0; u < cnt; ++u) acc = __dadd_rn(acc, t[u]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 12) {
        const int c = threadIdx.x;
        double* s = sums_all + (long long)pair * 15;
        if (c < 6) { s[fa[c] * 3 + fb[c]] = acc; s[fb[c] * 3 + fa[c]] = acc; }
        else s[9 + (c - 6)] = acc;
    }
}

// ---------------------------------------------------------------------------
// motion.compensate_frame (motion.py:289-321) fused with the squared error against
// `cur` (utils.py:109).  The field is either given (mf32) or evaluated from the affine
// parameters per block (results.py:52-54).
// ---------------------------------------------------------------------------
constexpr int COMP_ROWS = 32;      // rows per workgroup: one atomic per 256 x 32 tile

// Each thread owns 4 horizontally adjacent pixels of COMP_ROWS/4 rows.  When the block size is a
// multiple of 4 the four pixels share one vector; if their sources lie inside the frame they
// are gathered with two aligned dword loads + v_alignbyte and stored as one dword.
__global__ void __launch_bounds__(256) k_compensate(const uint8_t* frames, long long frame_stride, int H, int W,
                                                     int pitch, const int32_t* mf32, const double* params, int h,
                                                     int w, uint8_t* out, long long out_stride, int out_pitch,
                                                     const uint8_t* cur, long long cur_stride,
                                                     unsigned long long* sse)
{
    __shared__ unsigned part[4];
    const int pair = blockIdx.z;
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int ybase = blockIdx.y * COMP_ROWS + (threadIdx.x >> 6);
    const uint8_t* f = frames + (long long)pair * frame_stride;
    const int bs = H / h;                               // height only, motion.py:303
    unsigned err = 0;
    if (x < W) {
        double p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0;
        if (!mf32) {
            const double* p = params + (long long)pair * 6;
            p0 = p[0]; p1 = p[1]; p2 = p[2]; p3 = p[3]; p4 = p[4]; p5 = p[5];
        }
        const bool quad = (bs & 3) == 0 && x + 4 <= W;  // 4 pixels, one block column
        const int j = x / bs;
        int last_i = -1, d0 = 0, d1 = 0;
        const int yend = min(H, (int)(blockIdx.y + 1) * COMP_ROWS);
        for (int y = ybase; y < yend; y += 4) {
            const int i = y / bs;
            uint8_t* o = out + (long long)pair * out_stride + (long long)y * out_pitch + x;
            const uint8_t* c = cur ? cur + (long long)pair * cur_stride + (long long)y * pitch + x : nullptr;
            if (quad) {
                uint32_t v = *(const uint32_t*)(f + (long long)y * pitch + x);
                if (i < h && j < w) {
                    if (i != last_i) {                  // the vector changes only at block-row borders
                        last_i = i;
                        if (mf32) {
                            const int32_t* m = mf32 + (((long long)pair * h + i) * w + j) * 2;
                            d0 = m[0]; d1 = m[1];
                        } else {
                            d0 = model_component(p0, p1, p2, i, j);
                            d1 = model_component(p3, p4, p5, i, j);
                        }
                    }
                    const long long sy = (long long)y - d1, sx = (long long)x - d0;
                    if (sy >= 0 && sy < H) {
                        if (sx >= 0 && sx + 4 <= W) {
                            const uint8_t* sp = f + sy * pitch + (sx & ~3ll);
                            const uint32_t lo = *(const uint32_t*)sp, hi = *(const uint32_t*)(sp + 4);
                            v = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)sx & 3u);
                        } else {                        // straddles the left/right frame edge: per pixel
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (sx + q >= 0 && sx + q < W)
                                    v = (v & ~(0xFFu << (8 * q))) | ((uint32_t)f[sy * pitch + sx + q] << (8 * q));
                        }
                    }
                }
                *(uint32_t*)o = v;
                if (c) {
                    const uint32_t cv = *(const uint32_t*)c;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int df = (int)((cv >> (8 * q)) & 0xFF) - (int)((v >> (8 * q)) & 0xFF);
                        err += (unsigned)(df * df);
                    }
                }
            } else {
                for (int q = 0; q < 4 && x + q < W; ++q) {
                    const int jq = (x + q) / bs;
                    uint8_t v = f[(long long)y * pitch + x + q];
                    if (i < h && jq < w) {
                        int e0, e1;
                        if (mf32) {
                            const int32_t* m = mf32 + (((long long)pair * h + i) * w + jq) * 2;
                            e0 = m[0]; e1 = m[1];
                        } else {
                            e0 = model_component(p0, p1, p2, i, jq);
                            e1 = model_component(p3, p4, p5, i, jq);
                        }
                        const long long sy = (long long)y - e1, sx = (long long)x + q - e0;
                        if (sy >= 0 && sx >= 0 && sy < H && sx < W) v = f[sy * pitch + sx];
                    }
                    o[q] = v;
                    if (c) { const int df = (int)c[q] - (int)v; err += (unsigned)(df * df); }
                }
            }
        }
    }
    if (sse) {                                          // integer sums: order independent, deterministic
        for (int m = 32; m > 0; m >>= 1) err += (unsigned)__shfl_xor((int)err, m, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = err;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long t = (unsigned long long)part[0] + part[1] + part[2] + part[3];
            if (t) atomicAdd(&sse[pair], t);
        }
    }
}

// Fast form for block sizes that are multiples of 16 on frames whose width is a multiple of 16
// (the GME pipeline: bs = 16 at 720x480 and 1920x1080): a thread owns 16 horizontally adjacent
// pixels -- one block column, so one vector -- of two rows 16 apart; the source run is fetched as
// five aligned dwords + four v_alignbyte, stored as one 16-byte vector, and the squared error
// against `cur` is three v_dot4_u32_u8 per dword (sum c^2 + sum v^2 - 2 sum c.v).  The untouched
// original is only read where the rule of motion.py:309-318 keeps it (source outside the frame,
// rows/columns beyond the field).  Workgroup tile: 256 x 32 pixels, one atomic.
// buffer resource over `bytes` bytes from a wave-uniform pointer: reads outside it return 0
__device__ __forceinline__ __amdgpu_buffer_rsrc_t span_rsrc(const uint8_t* p, int bytes)
{
    const uint64_t base = (uint64_t)p;
    return __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((uint64_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                (uint32_t)__builtin_amdgcn_readfirstlane((int)base)),
        (short)0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

__global__ void __launch_bounds__(256) k_compensate16(const uint8_t* frames, long long frame_stride, int H, int W,
                                                       int pitch, const int32_t* mf32, const double* params, int h,
                                                       int w, uint8_t* out, long long out_stride, int out_pitch,
                                                       const uint8_t* cur, long long cur_stride,
                                                       unsigned long long* sse)
{
    __shared__ unsigned part[4];
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const int pair = blockIdx.z;
    const int x = (blockIdx.x * 16 + (threadIdx.x & 15)) * 16;
    const int y0 = blockIdx.y * 32 + (threadIdx.x >> 4);
    const uint8_t* f = frames + (long long)pair * frame_stride;
    const int bs = H / h;                               // height only, motion.py:303
    unsigned err = 0;
    if (x < W) {
        const int j = x / bs;
        // Three passes over the thread's two rows so that all of its loads are in flight together: (1) where each run
        // comes from, (2) the loads, without a branch around them -- through buffer resources, a run that is not
        // fetched gets an out-of-range offset and reads 0 -- and (3) assembling, storing, squared error.
        enum { NONE, INSIDE, STRADDLE, KEEP };
        const int OUTSIDE = (int)0x80000000;
        int kind[2], soff[2];
        uint32_t sh[2];
        long long sxs[2], sys[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int y = y0 + 16 * half;
            kind[half] = y < H ? KEEP : NONE; soff[half] = OUTSIDE; sh[half] = 0; sxs[half] = sys[half] = 0;
            const int i = y / bs;
            if (y < H && i < h && j < w) {
                int d0, d1;
                if (mf32) {
                    const int32_t* m = mf32 + (((long long)pair * h + i) * w + j) * 2;
                    d0 = m[0]; d1 = m[1];
                } else {
                    const double* p = params + (long long)pair * 6;
                    d0 = model_component(p[0], p[1], p[2], i, j);
                    d1 = model_component(p[3], p[4], p[5], i, j);
                }
                const long long sy = (long long)y - d1, sx = (long long)x - d0;
                if (sy >= 0 && sy < H) {
                    if (sx >= 0 && sx + 16 <= W) {
                        kind[half] = INSIDE;
                        soff[half] = (int)(sy * pitch + (sx & ~3ll));
                        sh[half] = (uint32_t)sx & 3u;
                    } else if (sx > -16 && sx < W) {            // straddles the left/right frame edge: per pixel, below
                        kind[half] = STRADDLE; sxs[half] = sx; sys[half] = sy;
                    }
                }
            }
        }
        const __amdgpu_buffer_rsrc_t rf = span_rsrc(f, H * pitch);
        const __amdgpu_buffer_rsrc_t rc = span_rsrc(cur ? cur + (long long)pair * cur_stride : f, cur ? H * pitch : 0);
        u32x4_t t[2], c[2];
        uint32_t t4[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // five aligned dwords hold the 16 source bytes; the fifth only when the run is not dword aligned (it lies
            // inside the row then: W % 16 == 0)
            t[half] = __builtin_amdgcn_raw_buffer_load_b128(rf, soff[half], 0, 0);
            t4[half] = __builtin_amdgcn_raw_buffer_load_b32(rf, sh[half] ? soff[half] + 16 : OUTSIDE, 0, 0);
            c[half] = __builtin_amdgcn_raw_buffer_load_b128(rc, kind[half] != NONE ? (y0 + 16 * half) * pitch + x : OUTSIDE, 0, 0);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int y = y0 + 16 * half;
            if (kind[half] == NONE) continue;
            uint4 v;
            if (kind[half] == INSIDE) {
                v.x = __builtin_amdgcn_alignbyte(t[half].y, t[half].x, sh[half]); v.y = __builtin_amdgcn_alignbyte(t[half].z, t[half].y, sh[half]);
                v.z = __builtin_amdgcn_alignbyte(t[half].w, t[half].z, sh[half]); v.w = __builtin_amdgcn_alignbyte(t4[half], t[half].w, sh[half]);
            } else {
                v = *(const uint4*)(f + (long long)y * pitch + x);      // the untouched original (motion.py:309-318)
                if (kind[half] == STRADDLE) {
                    uint32_t vv[4] = { v.x, v.y, v.z, v.w };
                    const long long sx = sxs[half], sy = sys[half];
                    for (int q = 0; q < 16; ++q)
                        if (sx + q >= 0 && sx + q < W)
                            vv[q >> 2] = (vv[q >> 2] & ~(0xFFu << (8 * (q & 3)))) | ((uint32_t)f[sy * pitch + sx + q] << (8 * (q & 3)));
                    v = make_uint4(vv[0], vv[1], vv[2], vv[3]);
                }
            }
            *(uint4*)(out + (long long)pair * out_stride + (long long)y * out_pitch + x) = v;
            if (cur) {
                const uint32_t cc[4] = { c[half].x, c[half].y, c[half].z, c[half].w }, vv[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned s2 = __builtin_amdgcn_udot4(cc[q], cc[q], __builtin_amdgcn_udot4(vv[q], vv[q], 0u, false), false);
                    err += s2 - 2u * __builtin_amdgcn_udot4(cc[q], vv[q], 0u, false);
                }
            }
        }
    }
    if (sse) {                                          // integer sums: order independent, deterministic
        for (int m = 32; m > 0; m >>= 1) err += (unsigned)__shfl_xor((int)err, m, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = err;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long t = (unsigned long long)part[0] + part[1] + part[2] + part[3];
            if (t) atomicAdd(&sse[pair], t);
        }
    }
}

__global__ void __launch_bounds__(256) k_sse(const uint8_t* a, long long a_stride, int a_pitch, const uint8_t* b,
                                              long long b_stride, int b_pitch, int H, int W,
                                              unsigned long long* sse)
{
    const int pair = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    unsigned err = 0;
    if (x < W && y < H) {
        const int df = (int)a[(long long)pair * a_stride + (long long)y * a_pitch + x] -
                       (int)b[(long long)pair * b_stride + (long long)y * b_pitch + x];
        err = (unsigned)(df * df);
    }
    for (int m = 32; m > 0; m >>= 1) err += (unsigned)__shfl_xor((int)err, m, 64);
    if ((threadIdx.x & 63) == 0 && err) atomicAdd(&sse[pair], (unsigned long long)err);
}

// Tight host frames (row stride W) arrive as ONE linear copy per chunk -- the form the DMA engines move at link
// speed; a 2-D copy with 720-byte rows reached 37 GB/s -- and are spread into the pitched planes here.
__global__ void __launch_bounds__(256) k_repack(const uint8_t* src, int H, int W, uint8_t* dst, int pitch, long long dst_stride)
{
    const int f = blockIdx.z, y = blockIdx.y;
    const uint8_t* s = src + ((long long)f * H + y) * W;
    uint8_t* o = dst + (long long)f * dst_stride + (long long)y * pitch;
    if ((W & 15) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && (pitch & 15) == 0 && (dst_stride & 15) == 0) {
        for (int x = (blockIdx.x * 256 + threadIdx.x) * 16; x < W; x += gridDim.x * 256 * 16)
            *(uint4*)(o + x) = *(const uint4*)(s + x);
    } else {
        for (int x = blockIdx.x * 256 + threadIdx.x; x < W; x += gridDim.x * 256) o[x] = s[x];
    }
}

}  // namespace

int launch_repack(gme_ctx* ctx, hipStream_t stream, const uint8_t* src, int count, int H, int W, uint8_t* dst, int pitch,
                  long long dst_stride)
{
    const int step = max_grid_planes();
    for (int first = 0; first < count; first += step) {
        const int n = count - first < step ? count - first : step;
        const int per_row = (W & 15) == 0 ? (W / 16 + 255) / 256 : (W + 255) / 256;
        hipLaunchKernelGGL(k_repack, dim3(per_row, H, n), dim3(256), 0, stream, src + (long long)first * H * W, H, W,
                           dst + (long long)first * dst_stride, pitch, dst_stride);
    }
    GME_HIP_TRY(hipGetLastError());
    return GME_OK;
}

// Planes / pairs one launch may put in grid.y or grid.z (hardware limit 65535).  GME_MAX_GRID_PAIRS
// lowers it so that tests reach the chunked paths with a handful of pairs.
int max_grid_planes()
{
    if (const char* e = getenv("GME_MAX_GRID_PAIRS")) { const int v = atoi(e); if (v >= 1 && v <= 32768) return v; }
    return 32768;
}

int launch_pyrdown(gme_ctx* ctx, const Plane& src, const Plane& dst)
{
    GME_REQUIRE(dst.H == (src.H + 1) / 2 && dst.W == (src.W + 1) / 2 && dst.count == src.count, GME_ERR_ARG,
                "pyrdown: destination shape mismatch");
    // host copy of pyr_interior_end
    int x_end = (((src.W - 12) / 2 + 4) < dst.W ? ((src.W - 12) / 2 + 4) : dst.W) & ~3;
    if (x_end < 4) x_end = 4;
    const int interior_quads = (x_end - 4) / 4;
    const int per_row = (dst.W < 4 ? dst.W : 4) + (dst.W - x_end > 0 ? dst.W - x_end : 0);
    const int step = max_grid_planes();
    // LDS-tiled form: widths that are multiples of 8 (output quads, 8-byte LDS reads) and fit the 64 KB of a workgroup
    const int lpitch = (PYR_APRON + src.W + 2 + 7) & ~7;
    const size_t lds = (size_t)PYR_ROWS * lpitch;
    const int pyr_per_row = (src.W + 15) / 16, pyr_quads = dst.W / 4;
    if (src.W % 8 == 0 && src.W >= 8 && src.pitch % 16 == 0 && lds <= 64 * 1024 && PYR_ROWS * pyr_per_row < 4096 && pyr_per_row < 256 &&
        (PYR_T / 2) * pyr_quads < 4096 && pyr_quads < 256 && !getenv("GME_FORCE_GENERIC") && !getenv("GME_PYR_NOLDS")) {
        const uint32_t magic_row = (1u << 20) / (uint32_t)pyr_per_row + 1u, magic_quads = (1u << 20) / (uint32_t)pyr_quads + 1u;
        for (int first = 0; first < src.count; first += step) {
            const int n = src.count - first < step ? src.count - first : step;
            hipLaunchKernelGGL(k_pyrdown_lds, dim3(1, (dst.H + PYR_T - 1) / PYR_T, n), dim3(256), lds, ctx->stream, src.at(first),
                               (long long)src.stride, src.H, src.W, src.pitch, dst.at(first), (long long)dst.stride, dst.H, dst.W,
                               dst.pitch, lpitch, magic_row, magic_quads);
        }
        GME_HIP_TRY(hipGetLastError());
        return GME_OK;
    }
    for (int first = 0; first < src.count; first += step) {            // grid.z holds at most 65535 planes
        const int n = src.count - first < step ? src.count - first : step;
        if (interior_quads > 0) {
            const dim3 grid((interior_quads + 63) / 64, (dst.H + 7) / 8, n);      // 4 waves x 2 output rows per workgroup
            hipLaunchKernelGGL(k_pyrdown, grid, dim3(256), 0, ctx->stream, src.at(first), (long long)src.stride, src.H, src.W,
                               src.pitch, dst.at(first), (long long)dst.stride, dst.H, dst.W, dst.pitch);
        }
        const dim3 egrid((dst.H * per_row + 255) / 256, 1, n);
        if (src.W >= 16 && src.W % 4 == 0 && !getenv("GME_FORCE_GENERIC"))
            hipLaunchKernelGGL(k_pyrdown_edge16, egrid, dim3(256), 0, ctx->stream, src.at(first), (long long)src.stride, src.H, src.W,
                               src.pitch, dst.at(first), (long long)dst.stride, dst.H, dst.W, dst.pitch);
        else
        hipLaunchKernelGGL(k_pyrdown_edge, egrid, dim3(256), 0, ctx->stream, src.at(first), (long long)src.stride, src.H, src.W,
                           src.pitch, dst.at(first), (long long)dst.stride, dst.H, dst.W, dst.pitch);
    }
    GME_HIP_TRY(hipGetLastError());
    return GME_OK;
}

int launch_first_params(gme_ctx* ctx, const int32_t* dense, int pairs, int n_blocks, float* params0)
{
    if (pairs == 0) return GME_OK;
    hipLaunchKernelGGL(k_first_params, dim3(pairs), dim3(256), 0, ctx->stream, dense, n_blocks, params0);
    GME_HIP_TRY(hipGetLastError());
    return GME_OK;
}

int launch_project_first(gme_ctx* ctx, const float* params0, int pairs, double* params_in)
{
    if (pairs == 0) return GME_OK;
    hipLaunchKernelGGL(k_project_first, dim3((pairs * 6 + 255) / 256), dim3(256), 0, ctx->stream, params0, pairs * 6, params_in);
    GME_HIP_TRY(hipGetLastError());
    return GME_OK;
}

int launch_mv_summary(gme_ctx* ctx, const int32_t* mf, int pairs, int n_blocks, double* rows)
{
    if (pairs == 0) return GME_OK;
    // 64 KiB of histogram + 32 bytes of static LDS per workgroup: fine on gfx950 (160 KiB), beyond the 64 KiB of other gfx9
    // parts the Makefile's ARCH could name -- refuse clearly instead of failing at launch
    GME_REQUIRE((size_t)ctx->prop.sharedMemPerBlock >= SUMMARY_BINS * sizeof(unsigned int) + 64, GME_ERR_ARG,
                "gme_seq_mv_summary needs %zu bytes of LDS per workgroup, this device offers %zu",
                SUMMARY_BINS * sizeof(unsigned int) + 64, (size_t)ctx->prop.sharedMemPerBlock);
    hipLaunchKernelGGL(k_mv_summary, dim3(pairs), dim3(256), SUMMARY_BINS * sizeof(unsigned int), ctx->stream, mf, n_blocks, rows);
    GME_HIP_TRY(hipGetLastError());
    return GME_OK;
}

int launch_affine_field(gme_ctx* ctx, const double* params, int pairs, int h, int w, int16_t* out)
{
    if (pairs == 0 || h * w == 0) return GME_OK;
    const int step = max_grid_planes();
    for (int first = 0; first < pairs; first += step) {
        const int n = pairs - first < step ? pairs - first : step;
        hipLaunchKernelGGL(k_affine_field, dim3((h * w + 255) / 256, n), dim3(256), 0, ctx->stream, params + (size_t)first * 6,
                           h, w, out + (size_t)first * h * w * 2);
    }
    GME_HIP_TRY(hipGetLastError());
    return GME_OK;
}

int launch_fit_level(gme_ctx* ctx, const int32_t* gt, int pairs, int h, int w, const double* params, int drop,
                     int level_H, int level_W, int16_t* model, uint8_t* mask, int32_t* diff, int32_t* thr,
                     double* sums, void* list)
{
    if (pairs == 0) return GME_OK;
    const double wgt = 1.0 / ((double)level_H * (double)level_W);      // motion.py:250
    // inlier list: LDS when it fits beside a second resident workgroup, else the global buffer
    const size_t need = (size_t)h * w * sizeof(int4);
    const int in_lds = need <= 40 * 1024;
    hipLaunchKernelGGL(k_fit_level, dim3(pairs), dim3(256), in_lds ? need : 0, ctx->stream, gt, h, w, params, drop, wgt,
                       model, mask, diff, thr, sums, (int4*)list, in_lds);
    GME_HIP_TRY(hipGetLastError());
    return GME_OK;
}

int launch_compensate(gme_ctx* ctx, const uint8_t* frames, int64_t frame_stride, int pairs, int H, int W, int pitch,
                      const int32_t* mf32, const double* params, int h, int w, uint8_t* out, int64_t out_stride,
                      int out_pitch, const uint8_t* cur, int64_t cur_stride, unsigned long long* sse)
{
    if (pairs == 0) return GME_OK;
    const int step = max_grid_planes();
    if (pairs > step) {                                                 // grid.z holds at most 65535 pairs
        for (int first = 0; first < pairs; first += step) {
            const int n = pairs - first < step ? pairs - first : step;
            const int rc = launch_compensate(ctx, frames + (int64_t)first * frame_stride, frame_stride, n, H, W, pitch,
                                             mf32 ? mf32 + (size_t)first * h * w * 2 : nullptr, params ? params + (size_t)first * 6 : nullptr,
                                             h, w, out + (int64_t)first * out_stride, out_stride, out_pitch,
                                             cur ? cur + (int64_t)first * cur_stride : nullptr, cur_stride, sse ? sse + first : nullptr);
            if (rc) return rc;
        }
        return GME_OK;
    }
    if (sse) GME_HIP_TRY(hipMemsetAsync(sse, 0, sizeof(unsigned long long) * pairs, ctx->stream));
    const dim3 grid((W + 255) / 256, (H + COMP_ROWS - 1) / COMP_ROWS, pairs);
    if (h > 0 && (H / h) % 16 == 0 && W % 16 == 0 && pitch % 16 == 0 && out_pitch % 16 == 0 && frame_stride % 16 == 0 &&
        out_stride % 16 == 0 && (!cur || cur_stride % 16 == 0) &&
        (((uintptr_t)frames | (uintptr_t)out | (uintptr_t)cur) & 15) == 0 && !getenv("GME_FORCE_GENERIC")) {
        hipLaunchKernelGGL(k_compensate16, grid, dim3(256), 0, ctx->stream, frames, (long long)frame_stride, H, W, pitch,
                           mf32, params, h, w, out, (long long)out_stride, out_pitch, cur, (long long)cur_stride, sse);
        GME_HIP_TRY(hipGetLastError());
        return GME_OK;
    }
    hipLaunchKernelGGL(k_compensate, grid, dim3(256), 0, ctx->stream, frames, (long long)frame_stride, H, W, pitch,
                       mf32, params, h, w, out, (long long)out_stride, out_pitch, cur, (long long)cur_stride, sse);
    GME_HIP_TRY(hipGetLastError());
    return GME_OK;
}

int launch_sse(gme_ctx* ctx, const uint8_t* a, int64_t a_stride, int a_pitch, const uint8_t* b, int64_t b_stride,
               int b_pitch, int pairs, int H, int W, unsigned long long* sse)
{
    if (pairs == 0) return GME_OK;
    GME_HIP_TRY(hipMemsetAsync(sse, 0, sizeof(unsigned long long) * pairs, ctx->stream));
    const int step = max_grid_planes();
    for (int first = 0; first < pairs; first += step) {
        const int n = pairs - first < step ? pairs - first : step;
        const dim3 grid((W + 63) / 64, (H + 3) / 4, n);
        hipLaunchKernelGGL(k_sse, grid, dim3(256), 0, ctx->stream, a + (int64_t)first * a_stride, (long long)a_stride, a_pitch,
                           b + (int64_t)first * b_stride, (long long)b_stride, b_pitch, H, W, sse + first);
    }
    GME_HIP_TRY(hipGetLastError());
    return GME_OK;
}

int launch_solve3(gme_ctx* ctx, const double* sums, int pairs, int project, int h, int w, double* params_out, int32_t* flags, int flag_bit)
{
    if (pairs == 0) return GME_OK;
    hipLaunchKernelGGL(k_solve3, dim3((unsigned)pairs), dim3(256), 0, ctx->stream, sums, pairs, project, h, w, 1e-9, params_out, flags, flag_bit);
    GME_HIP_TRY(hipGetLastError());
    return GME_OK;
}
