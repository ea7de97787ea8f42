// Block-matching motion estimation kernels for gfx950 (MI355X).
//
// Replaces bbme.py:105-534 of the reference (exhaustive, three-step, 2-D log and
// diamond searches under MAE / MSE).  All block distances are exact integers, which
// equals the reference's float32 arithmetic inside the bounds checked by
// bbme_check_args() (bbme.py:61-64; SURVEY.md §7 hard part 3).
//
// Kernels
//   k_exh_qsad16<R>   exhaustive MAE, bs = 16: one wavefront per macroblock, the
//                     anchor block in SGPRs, the search window staged in LDS, the
//                     4-offset sliding SAD instruction v_qsad_pk_u16_u8 in the loop,
//                     packed (cost, scan index) keys min-reduced across the wave.
//   k_exh_generic     exhaustive, any geometry / norm (one 256-thread group per block)
//   k_walk<G>         three-step / 2-D log / diamond walks, G lanes per block
#include <stdlib.h>

#include "gme_internal.h"

namespace {

struct Dev {
    const uint8_t* prev;
    const uint8_t* cur;
    long long plane_stride;
    int pairs, H, W, pitch, bs, sw, pnorm, procedure;
    int nbr, nbc;
    int32_t* mf;
    int* status;
    int f32;                      // block distances in NumPy's float32 pairwise order (see block_cost_f32)
};

// ---------------------------------------------------------------------------
// Block distance exactly as the reference computes it when the sum leaves float32's
// exact-integer range (MSE with bs > 16, MAE with bs > 256): compute_dfd (bbme.py:41-64,79,94)
// hands a contiguous float32 bs x bs array to np.sum, which reduces the flat n = bs*bs
// elements with NumPy's pairwise summation -- 8 strided accumulators over blocks of <= 128
// elements, halves split at multiples of 8 (numpy/core/src/umath/loops_utils.h.src).  The
// emulation below reproduces that order bit for bit (checked against np.sum under NumPy
// 1.26.4 and 2.2.6 in tests/test_host.py).  Non-negative floats order like their bit patterns.
// ---------------------------------------------------------------------------
struct BlockPair {
    const uint8_t* a;
    const uint8_t* c;
    int pitch, bs, pnorm;
    __device__ __forceinline__ float operator()(int idx) const
    {
        const int y = idx / bs, x = idx - y * bs;
        const float df = (float)a[y * pitch + x] - (float)c[y * pitch + x];
        return pnorm ? __fmul_rn(df, df) : fabsf(df);
    }
};

__device__ float pairwise_f32(const BlockPair& e, int start, int n)
{
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res = __fadd_rn(res, e(start + i));
        return res;
    }
    if (n <= 128) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = e(start + j);
        int i = 8;
        for (; i < n - (n % 8); i += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], e(start + i + j));
        float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                              __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
        for (; i < n; ++i) res = __fadd_rn(res, e(start + i));
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return __fadd_rn(pairwise_f32(e, start, n2), pairwise_f32(e, start + n2, n - n2));
}

__device__ __forceinline__ unsigned long long block_cost_f32(const uint8_t* a, const uint8_t* c, int pitch, int bs,
                                                             int pnorm)
{
    const BlockPair e{ a, c, pitch, bs, pnorm };
    return (unsigned long long)__float_as_uint(pairwise_f32(e, 0, bs * bs));
}

__device__ __forceinline__ unsigned long long u64min(unsigned long long a, unsigned long long b)
{
    return a < b ? a : b;
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// ---------------------------------------------------------------------------
// generic exhaustive search (bbme.py:105-179)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_exh_generic(Dev d)
{
    __shared__ unsigned long long wave_best[4];
    const int nblk = d.nbr * d.nbc;
    const int pair = blockIdx.x / nblk, blk = blockIdx.x % nblk;
    const int r0 = (blk / d.nbc) * d.bs, c0 = (blk % d.nbc) * d.bs;
    const uint8_t* prev = d.prev + (long long)pair * d.plane_stride + (long long)r0 * d.pitch + c0;
    const uint8_t* cur = d.cur + (long long)pair * d.plane_stride;
    const int nc = 2 * d.sw + d.bs;
    const int total = nc * nc;
    unsigned long long best = ~0ull;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int ci = idx / nc, ri = idx - ci * nc;        // column offset is the outer loop
        const int top = r0 + ri - d.sw, left = c0 + ci - d.sw;
        if (top < 0 || left < 0 || top + d.bs > d.H || left + d.bs > d.W) continue;
        const uint8_t* cand = cur + (long long)top * d.pitch + left;
        unsigned long long cost = 0;
        if (d.f32) cost = block_cost_f32(prev, cand, d.pitch, d.bs, d.pnorm);
        else for (int y = 0; y < d.bs; ++y) {
            const uint8_t* a = prev + (long long)y * d.pitch;
            const uint8_t* c = cand + (long long)y * d.pitch;
            unsigned row = 0;
            if (d.pnorm == 0) {
                for (int x = 0; x < d.bs; ++x) row += (unsigned)abs((int)a[x] - (int)c[x]);
                cost += row;
            } else {
                unsigned long long r64 = 0;
                for (int x = 0; x < d.bs; ++x) { int df = (int)a[x] - (int)c[x]; r64 += (unsigned)(df * df); }
                cost += r64;
            }
        }
        best = u64min(best, (cost << 24) | (unsigned)idx);   // first minimum in scan order
    }
    for (int m = 32; m > 0; m >>= 1) best = u64min(best, shfl_xor_u64(best, m));
    if ((threadIdx.x & 63) == 0) wave_best[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        best = u64min(u64min(wave_best[0], wave_best[1]), u64min(wave_best[2], wave_best[3]));
        const int idx = (int)(best & 0xFFFFFF);
        const int ci = idx / nc, ri = idx - ci * nc;
        int32_t* o = d.mf + ((long long)pair * nblk + blk) * 2;
        o[0] = ci - d.sw;
        o[1] = ri - d.sw;
    }
}

// ---------------------------------------------------------------------------
// walk searches: G lanes cooperate on one block
// ---------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ unsigned long long group_cost(const uint8_t* anchor, const uint8_t* cand,
                                                         int pitch, int bs, int pnorm, int lig, int f32)
{
    if (G == 1 && f32) return block_cost_f32(anchor, cand, pitch, bs, pnorm);
    unsigned long long acc = 0;
    const int n = bs * bs;
    int y = lig / bs, x = lig - y * bs;
    const int dy = G / bs, dx = G - dy * bs;
    for (int p = lig; p < n; p += G) {
        const int df = (int)anchor[y * pitch + x] - (int)cand[y * pitch + x];
        acc += pnorm ? (unsigned)(df * df) : (unsigned)abs(df);
        y += dy; x += dx;
        if (x >= bs) { x -= bs; ++y; }
    }
#pragma unroll
    for (int m = G / 2; m > 0; m >>= 1) acc += shfl_xor_u64(acc, m);
    return acc;
}

__constant__ int c_ldsp[9][2] = { {0,0},{2,0},{1,1},{0,2},{-1,1},{-2,0},{-1,-1},{0,-2},{1,-1} };
__constant__ int c_sdsp[5][2] = { {0,0},{1,0},{0,1},{-1,0},{0,-1} };

__device__ __forceinline__ int clamp_ref(int v, int hi)   // min(max(v, 0), hi), bbme.py:503-504
{
    const int t = v > 0 ? v : 0;
    return t < hi ? t : hi;
}

template <int G>
__global__ void __launch_bounds__(256) k_walk(Dev d)
{
    const int nblk = d.nbr * d.nbc;
    const long long total = (long long)nblk * d.pairs;
    const long long gid = ((long long)blockIdx.x * 256 + threadIdx.x) / G;
    const int lig = threadIdx.x % G;
    const bool active = gid < total;
    const long long g = active ? gid : 0;
    const int pair = (int)(g / nblk), blk = (int)(g % nblk);
    const int r0 = (blk / d.nbc) * d.bs, c0 = (blk % d.nbc) * d.bs;
    const uint8_t* cur = d.cur + (long long)pair * d.plane_stride;
    const uint8_t* anchor = d.prev + (long long)pair * d.plane_stride + (long long)r0 * d.pitch + c0;
    const int bs = d.bs, H = d.H, W = d.W, pitch = d.pitch, pnorm = d.pnorm;
    const unsigned long long INF = ~0ull;
    const int cap = 2 * (H + W) + 64;       // walks strictly decrease the cost; this is a guard
    bool overrun = false;
    int out0 = 0, out1 = 0;

#define COST_AT(rr, cc) group_cost<G>(anchor, cur + (long long)(rr) * pitch + (cc), pitch, bs, pnorm, lig, d.f32)
#define INSIDE(rr, cc) ((rr) >= 0 && (cc) >= 0 && (rr) + bs <= H && (cc) + bs <= W)

    if (d.procedure == GME_SEARCH_DIAMOND) {           // bbme.py:436-534
        const int maxr = H - bs - 1, maxc = W - bs - 1;
        int pr = r0, pc = c0, br = r0, bc = c0;
        bool done = !active;
        int it = 0;
        while (__any(!done)) {
            unsigned long long best = INF;
            int nr = pr, nc = pc;
            for (int k = 0; k < 9; ++k) {
                const int rr = clamp_ref(pr + c_ldsp[k][0], maxr), cc = clamp_ref(pc + c_ldsp[k][1], maxc);
                const unsigned long long c = COST_AT(rr, cc);
                if (c < best) { best = c; nr = rr; nc = cc; }
            }
            if (!done) {
                done = (nr == pr && nc == pc);
                pr = nr; pc = nc;
            }
            if (++it > cap) { overrun = true; break; }
        }
        unsigned long long best = INF;
        br = pr; bc = pc;
        for (int k = 0; k < 5; ++k) {                  // offsets applied swapped, bbme.py:518-521
            const int rr = clamp_ref(pr + c_sdsp[k][1], maxr), cc = clamp_ref(pc + c_sdsp[k][0], maxc);
            const unsigned long long c = COST_AT(rr, cc);
            if (c < best) { best = c; br = rr; bc = cc; }
        }
        out1 = br - r0; out0 = bc - c0;
    } else if (d.procedure == GME_SEARCH_THREESTEP) {  // bbme.py:182-341
        const int n = 2 * d.sw + bs;
        const int steps[3] = { (int)(n / 3.0), (int)(n / 5.0), (int)(n / 10.0) };
        int drow = 0, dcol = 0, trow = 0, tcol = 0;
        int org_r = r0, org_c = c0;
        for (int s = 0; s < 3; ++s) {
            const int st = steps[s];
            unsigned long long best = INF;
            int kr = s == 0 ? drow : trow, kc = s == 0 ? dcol : tcol;
            for (int a = -1; a <= 1; ++a)
                for (int b = -1; b <= 1; ++b) {
                    const int wc = a * st, wr = b * st;
                    const int rr = org_r + wr, cc = org_c + wc;
                    // every lane of the wave must take part in the shuffles of COST_AT
                    const bool ok = INSIDE(rr, cc);
                    const unsigned long long c = COST_AT(ok ? rr : r0, ok ? cc : c0);
                    if (ok && c < best) { best = c; kr = wr; kc = wc; }
                }
            if (s == 0) { drow = kr; dcol = kc; org_r = r0 + drow; org_c = c0 + dcol; }
            else {
                trow = kr; tcol = kc;
                drow += trow; dcol += tcol;
                org_r += drow; org_c += dcol;          // bbme.py:300-301 (accumulated again)
            }
        }
        out0 = dcol; out1 = drow;
    } else {                                           // 2-D log, bbme.py:344-433
        int br = 0, bc = 0, pr = r0, pc = c0;
        int step = active ? d.sw : 0;
        int it = 0;
        while (__any(step > 1)) {
            const bool cross = step > 2;
            unsigned long long best = INF;
            int nr = br, nc = bc;
            for (int k = 0; k < 9; ++k) {
                int rr, cc;
                if (cross) {
                    if (k >= 5) break;
                    rr = pr + (k == 1 ? step : k == 2 ? -step : 0);
                    cc = pc + (k == 3 ? step : k == 4 ? -step : 0);
                } else {
                    rr = pr + (k / 3 - 1) * 2;
                    cc = pc + (k % 3 - 1) * 2;
                }
                const bool ok = INSIDE(rr, cc);
                const unsigned long long c = COST_AT(ok ? rr : r0, ok ? cc : c0);
                if (ok && c < best) { best = c; nr = rr; nc = cc; }
            }
            if (step > 1) {
                br = nr; bc = nc;
                if ((br == pr && bc == pc) || step == 2) step /= 2;
                pr = br; pc = bc;
            }
            if (++it > cap) { overrun = true; break; }
        }
        out1 = br - r0; out0 = bc - c0;
    }
#undef COST_AT
#undef INSIDE
    if (overrun && lig == 0) atomicExch(d.status, 1);
    if (active && lig == 0) {
        int32_t* o = d.mf + gid * 2;
        o[0] = out0; o[1] = out1;
    }
}


// ---------------------------------------------------------------------------
// Walk searches for block sizes 4, 8, 12, 20 .. 32 (multiples of 4 other than 16, costs inside float32's exact-integer
// range): the sizes the reference itself runs besides 16 -- get_motion_field's default block_size=4 (bbme.py:15-18) and
// the authors' BBME_BLOCK_SIZE = 12 / 24 / 32 (motion.py:9, docs/presentation/main.tex:382,426,558).  Same control flow as
// k_walk<G> above (candidate order, clamps, first strict minimum: bbme.py:182-534), a different block distance:
//   * G = 1 / 2 / 4 / 8 lanes per block, each lane owning whole block ROWS (lig, lig + G, ...): its anchor rows live in
//     registers for the whole walk (BS / 4 dwords each);
//   * a candidate row is BS / 4 + 1 aligned dwords of `cur` (the frames of a pair are L2-resident) turned into the
//     BS / 4 unaligned ones by v_alignbyte_b32, then one v_sad_u8 (MAE) or three v_dot4_u32_u8 (MSE:
//     sum a^2 + sum b^2 - 2 sum ab) per dword -- k_walk<G> reads single bytes and accumulates in 64 bits;
//   * the G partial costs are added with DPP moves inside the group (32-bit: BS^2 * 65025 < 2^32 for BS <= 32).
// A 16-byte row read may run up to 3 bytes past the block's right edge: inside the pitch padding, or into the next row /
// the guard row behind every plane (plane_alloc) -- those bytes are shifted out by v_alignbyte.
// ---------------------------------------------------------------------------
template <int BS> struct WalkQ {
    static constexpr int G = BS <= 4 ? 1 : BS <= 8 ? 2 : BS <= 16 ? 4 : 8;      // lanes per block
    static constexpr int RPL = (BS + G - 1) / G;                                 // block rows per lane
    static constexpr int DW = BS / 4;                                            // dwords per block row
};

template <int G>
__device__ __forceinline__ unsigned groupq_sum(unsigned v)
{
    if (G >= 2) v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    if (G >= 4) v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    if (G >= 8) v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    return v;
}

template <int BS, int PNORM>
__global__ void __launch_bounds__(256) k_walkq(Dev d)
{
    constexpr int G = WalkQ<BS>::G, RPL = WalkQ<BS>::RPL, DW = WalkQ<BS>::DW;
    const int nblk = d.nbr * d.nbc;
    const long long total = (long long)nblk * d.pairs;
    const long long gid = ((long long)blockIdx.x * 256 + threadIdx.x) / G;
    const int lig = threadIdx.x % G;
    const bool active = gid < total;
    const long long g = active ? gid : 0;
    const int pair = (int)(g / nblk), blk = (int)(g % nblk);
    const int r0 = (blk / d.nbc) * BS, c0 = (blk % d.nbc) * BS;
    const uint8_t* cur = d.cur + (long long)pair * d.plane_stride;
    const uint8_t* anchor = d.prev + (long long)pair * d.plane_stride + (long long)r0 * d.pitch + c0;
    const int H = d.H, W = d.W, pitch = d.pitch;
    const unsigned INF = 0xFFFFFFFFu;
    const int cap = 2 * (H + W) + 64;       // walks strictly decrease the cost; this is a guard
    bool overrun = false;
    int out0 = 0, out1 = 0;

    // this lane's anchor rows (block origins are multiples of 4: aligned dwords) and, for MSE, their sum of squares
    uint32_t a[RPL][DW];
    unsigned aa = 0;
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
        const int row = lig + G * j;
#pragma unroll
        for (int k = 0; k < DW; ++k) {
            a[j][k] = (BS % G == 0 || row < BS) ? *(const uint32_t*)(anchor + (long long)row * pitch + 4 * k) : 0u;
            if (PNORM == 1) aa = __builtin_amdgcn_udot4(a[j][k], a[j][k], aa, false);
        }
    }

    auto cost_at = [&](int rr, int cc) -> unsigned {
        const uint8_t* p = cur + (long long)(rr + lig) * pitch + (cc & ~3);
        const uint32_t sh = (uint32_t)cc & 3u;
        unsigned part = PNORM == 1 ? aa : 0u, ab = 0;
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
            if (BS % G != 0 && lig + G * j >= BS) break;                 // ragged last row group (BS = 12 with G = 8 never happens; BS = 20, 28)
            uint32_t l[DW + 1];
#pragma unroll
            for (int k = 0; k <= DW; ++k) l[k] = *(const uint32_t*)(p + (long long)(G * j) * pitch + 4 * k);
#pragma unroll
            for (int k = 0; k < DW; ++k) {
                const uint32_t b = __builtin_amdgcn_alignbyte(l[k + 1], l[k], sh);
                if (PNORM == 0) part = __builtin_amdgcn_sad_u8(a[j][k], b, part);
                else {
                    part = __builtin_amdgcn_udot4(b, b, part, false);
                    ab = __builtin_amdgcn_udot4(a[j][k], b, ab, false);
                }
            }
        }
        if (PNORM == 1) part -= 2u * ab;                                  // this lane's rows: sum (a - b)^2 >= 0
        return groupq_sum<G>(part);
    };
#define INSIDE(rr, cc) ((rr) >= 0 && (cc) >= 0 && (rr) + BS <= H && (cc) + BS <= W)

    if (d.procedure == GME_SEARCH_DIAMOND) {           // bbme.py:436-534
        const int maxr = H - BS - 1, maxc = W - BS - 1;
        int pr = r0, pc = c0, br = r0, bc = c0;
        bool done = !active;
        int it = 0;
        while (__any(!done)) {
            unsigned best = INF;
            int nr = pr, nc = pc;
            for (int k = 0; k < 9; ++k) {
                const int rr = clamp_ref(pr + c_ldsp[k][0], maxr), cc = clamp_ref(pc + c_ldsp[k][1], maxc);
                const unsigned c = cost_at(rr, cc);
                if (c < best) { best = c; nr = rr; nc = cc; }
            }
            if (!done) {
                done = (nr == pr && nc == pc);
                pr = nr; pc = nc;
            }
            if (++it > cap) { overrun = true; break; }
        }
        unsigned best = INF;
        br = pr; bc = pc;
        for (int k = 0; k < 5; ++k) {                  // offsets applied swapped, bbme.py:518-521
            const int rr = clamp_ref(pr + c_sdsp[k][1], maxr), cc = clamp_ref(pc + c_sdsp[k][0], maxc);
            const unsigned c = cost_at(rr, cc);
            if (c < best) { best = c; br = rr; bc = cc; }
        }
        out1 = br - r0; out0 = bc - c0;
    } else if (d.procedure == GME_SEARCH_THREESTEP) {  // bbme.py:182-341
        const int n = 2 * d.sw + BS;
        const int steps[3] = { (int)(n / 3.0), (int)(n / 5.0), (int)(n / 10.0) };
        int drow = 0, dcol = 0, trow = 0, tcol = 0;
        int org_r = r0, org_c = c0;
        for (int s = 0; s < 3; ++s) {
            const int st = steps[s];
            unsigned best = INF;
            int kr = s == 0 ? drow : trow, kc = s == 0 ? dcol : tcol;
            for (int a2 = -1; a2 <= 1; ++a2)
                for (int b2 = -1; b2 <= 1; ++b2) {
                    const int wc = a2 * st, wr = b2 * st;
                    const int rr = org_r + wr, cc = org_c + wc;
                    // every lane of the wave must take part in the DPP moves of cost_at
                    const bool ok = INSIDE(rr, cc);
                    const unsigned c = cost_at(ok ? rr : r0, ok ? cc : c0);
                    if (ok && c < best) { best = c; kr = wr; kc = wc; }
                }
            if (s == 0) { drow = kr; dcol = kc; org_r = r0 + drow; org_c = c0 + dcol; }
            else {
                trow = kr; tcol = kc;
                drow += trow; dcol += tcol;
                org_r += drow; org_c += dcol;          // bbme.py:300-301 (accumulated again)
            }
        }
        out0 = dcol; out1 = drow;
    } else {                                           // 2-D log, bbme.py:344-433
        int br = 0, bc = 0, pr = r0, pc = c0;
        int step = active ? d.sw : 0;
        int it = 0;
        while (__any(step > 1)) {
            const bool cross = step > 2;
            unsigned best = INF;
            int nr = br, nc = bc;
            for (int k = 0; k < 9; ++k) {
                int rr, cc;
                if (cross) {
                    if (k >= 5) break;
                    rr = pr + (k == 1 ? step : k == 2 ? -step : 0);
                    cc = pc + (k == 3 ? step : k == 4 ? -step : 0);
                } else {
                    rr = pr + (k / 3 - 1) * 2;
                    cc = pc + (k % 3 - 1) * 2;
                }
                const bool ok = INSIDE(rr, cc);
                const unsigned c = cost_at(ok ? rr : r0, ok ? cc : c0);
                if (ok && c < best) { best = c; nr = rr; nc = cc; }
            }
            if (step > 1) {
                br = nr; bc = nc;
                if ((br == pr && bc == pc) || step == 2) step /= 2;
                pr = br; pc = bc;
            }
            if (++it > cap) { overrun = true; break; }
        }
        out1 = br - r0; out0 = bc - c0;
    }
#undef INSIDE
    if (overrun && lig == 0) atomicExch(d.status, 1);
    if (active && lig == 0) {
        int32_t* o = d.mf + gid * 2;
        o[0] = out0; o[1] = out1;
    }
}

}  // namespace

// the `break` inside the 2-D log candidate loop is wave-divergent only between groups
// whose `cross` differs; COST_AT shuffles stay inside a group (width G), and every lane
// of a group shares `cross`, so no shuffle partner is ever missing.

int bbme_check_args(int H, int W, int bs, int sw, int procedure, int pnorm)
{
    GME_REQUIRE(procedure >= 0 && procedure <= 3, GME_ERR_ARG, "searching_procedure %d out of range (bbme.py:27)", procedure);
    GME_REQUIRE(pnorm == 0 || pnorm == 1, GME_ERR_ARG, "pnorm_distance %d out of range (bbme.py:60)", pnorm);
    GME_REQUIRE(bs >= 1 && H >= 1 && W >= 1, GME_ERR_ARG, "bad geometry H=%d W=%d bs=%d", H, W, bs);
    GME_REQUIRE(bs <= 4096, GME_ERR_ARG, "block_size %d too large", bs);
    GME_REQUIRE((long long)bs * bs < (1ll << 30), GME_ERR_ARG, "block_size %d too large", bs);
    if (procedure == GME_SEARCH_EXHAUSTIVE) {
        GME_REQUIRE(sw >= 0 && 2 * sw + bs < 4096, GME_ERR_ARG, "search_window %d out of range", sw);
    }
    if (procedure == GME_SEARCH_DIAMOND && H / bs > 0 && W / bs > 0) {
        GME_REQUIRE(H - bs - 1 >= 0 && W - bs - 1 >= 0, GME_ERR_GEOMETRY,
                    "diamond search needs H > bs and W > bs (bbme.py:503-505 slices an empty block otherwise)");
    }
    return GME_OK;
}

int launch_bbme_fast(gme_ctx* ctx, const BbmeJob& job, bool* handled);
int launch_bbme_sea(gme_ctx* ctx, const BbmeJob& job, bool* handled);
int launch_bbme_sea_mse(gme_ctx* ctx, const BbmeJob& job, bool* handled);
int launch_bbme_walk_fast(gme_ctx* ctx, const BbmeJob& job, bool* handled);

// the reference's float32 sums stop being exact integers at 2^24 (bbme.py:61-64)
static bool needs_f32_order(int bs, int pnorm)
{
    return (double)bs * bs * (pnorm ? 65025.0 : 255.0) >= 16777216.0;
}

static int launch_bbme_chunk(gme_ctx* ctx, const BbmeJob& job);

int launch_bbme(gme_ctx* ctx, const BbmeJob& job)
{
    int rc = bbme_check_args(job.H, job.W, job.bs, job.sw, job.procedure, job.pnorm);
    if (rc != GME_OK) return rc;
    // long sequences: keep every launch below ~16 M blocks so grid sizes stay far from 2^31
    const long long nblk = (long long)(job.H / job.bs) * (job.W / job.bs);
    if (nblk == 0 || job.pairs == 0) return GME_OK;
    if (!job.chained) {
        ctx->plan[0] = 0;
        ctx->plan_patches = 0;
        // one fill for the statistics AND the first chunk's tile counters and redo words (three small fills cost ~2 us each
        // in front of a 3.7 ms launch)
        GME_HIP_TRY(hipMemsetAsync(ctx->status + GME_STATUS_TILECTR, 0, (GME_STATUS_REDO + 2 - GME_STATUS_TILECTR) * sizeof(int), ctx->stream));
    }
    long long cap = 1ll << 24;
    if (const char* e = getenv("GME_BBME_CHUNK_BLOCKS")) {             // test hook: reach the chunked path with few pairs
        const long long v = atoll(e);
        if (v >= 1 && v <= (1ll << 24)) cap = v;
    }
    const long long per = cap / nblk < 1 ? 1 : cap / nblk;
    for (long long first = 0; first < job.pairs; first += per) {
        BbmeJob part = job;
        part.status_fresh = first == 0 && !job.chained;
        part.pairs = (int)(job.pairs - first < per ? job.pairs - first : per);
        part.prev = job.prev + first * job.plane_stride;
        part.cur = job.cur + first * job.plane_stride;
        part.mf = job.mf + first * nblk * 2;
        if (job.sqbox_cur) part.sqbox_cur = job.sqbox_cur + first * job.sqbox_stride;
        rc = launch_bbme_chunk(ctx, part);
        if (rc != GME_OK) return rc;
    }
    return GME_OK;
}

static int launch_bbme_chunk(gme_ctx* ctx, const BbmeJob& job)
{
    int rc = GME_OK;
    Dev d;
    d.f32 = needs_f32_order(job.bs, job.pnorm) ? 1 : 0;
    d.prev = job.prev; d.cur = job.cur; d.plane_stride = job.plane_stride;
    d.pairs = job.pairs; d.H = job.H; d.W = job.W; d.pitch = job.pitch;
    d.bs = job.bs; d.sw = job.sw; d.pnorm = job.pnorm; d.procedure = job.procedure;
    d.nbr = job.H / job.bs; d.nbc = job.W / job.bs;
    d.mf = job.mf;
    const long long nblk = (long long)d.nbr * d.nbc;
    if (nblk == 0 || job.pairs == 0) return GME_OK;
    GME_REQUIRE(nblk * job.pairs < (1ll << 31) / 64, GME_ERR_ARG, "too many blocks in one launch");

    bool handled = false;
    rc = launch_bbme_mfma(ctx, job, &handled);              // exhaustive MSE at bs 16: the correlation on the matrix cores
    if (rc != GME_OK || handled) return rc;
    rc = launch_bbme_sea(ctx, job, &handled);
    if (rc != GME_OK || handled) return rc;
    rc = launch_bbme_sea_mse(ctx, job, &handled);
    if (rc != GME_OK || handled) return rc;
    rc = launch_bbme_fast(ctx, job, &handled);
    if (rc != GME_OK || handled) return rc;
    if (!getenv("GME_FORCE_GENERIC")) {
        rc = launch_bbme_walk_fast(ctx, job, &handled);
        if (rc != GME_OK || handled) return rc;
    }

    d.status = ctx->status;

    if (job.procedure == GME_SEARCH_EXHAUSTIVE) {
        plan_note(ctx, 0, "k_exh_generic%s grid %lld", d.f32 ? " (float32-order costs)" : "", nblk * job.pairs);
        hipLaunchKernelGGL(k_exh_generic, dim3((unsigned)(nblk * job.pairs)), dim3(256), 0, ctx->stream, d);
    } else if (!d.f32 && job.bs % 4 == 0 && job.bs >= 4 && job.bs <= 32 && job.bs != 16 && !getenv("GME_FORCE_GENERIC")) {
        // the reference's other block sizes (k_walkq above); 16 has k_walk16 / k_walk16s
        const int G = job.bs <= 4 ? 1 : job.bs <= 8 ? 2 : job.bs <= 16 ? 4 : 8;
        const long long threads = nblk * job.pairs * G;
        const unsigned grid = (unsigned)((threads + 255) / 256);
        plan_note(ctx, 0, "k_walkq<%d,%d> %d lane(s) per block grid %u", job.bs, job.pnorm, G, grid);
#define WALKQ(B) do { if (job.pnorm == 0) hipLaunchKernelGGL((k_walkq<B, 0>), dim3(grid), dim3(256), 0, ctx->stream, d); \
                      else hipLaunchKernelGGL((k_walkq<B, 1>), dim3(grid), dim3(256), 0, ctx->stream, d); } while (0)
        switch (job.bs) {
        case 4: WALKQ(4); break;
        case 8: WALKQ(8); break;
        case 12: WALKQ(12); break;
        case 20: WALKQ(20); break;
        case 24: WALKQ(24); break;
        case 28: WALKQ(28); break;
        default: WALKQ(32); break;
        }
#undef WALKQ
    } else {
        const int px = job.bs * job.bs;
        // float32-order costs are sequential by definition: one lane per block
        const int G = d.f32 ? 1 : px <= 4 ? 1 : px <= 16 ? 4 : px <= 64 ? 16 : 64;
        const long long threads = nblk * job.pairs * G;
        const unsigned grid = (unsigned)((threads + 255) / 256);
        plan_note(ctx, 0, "k_walk<%d>%s grid %u", G, d.f32 ? " (float32-order costs)" : "", grid);
        switch (G) {
        case 1: hipLaunchKernelGGL(k_walk<1>, dim3(grid), dim3(256), 0, ctx->stream, d); break;
        case 4: hipLaunchKernelGGL(k_walk<4>, dim3(grid), dim3(256), 0, ctx->stream, d); break;
        case 16: hipLaunchKernelGGL(k_walk<16>, dim3(grid), dim3(256), 0, ctx->stream, d); break;
        default: hipLaunchKernelGGL(k_walk<64>, dim3(grid), dim3(256), 0, ctx->stream, d); break;
        }
    }
    GME_HIP_TRY(hipGetLastError());
    return GME_OK;
}
