// Exhaustive MAE search, bs = 16, with exact successive elimination -- same answer as the brute
// force kernel (bbme_fast.hip: k_exh_qsad16, bbme.py:105-179), a fraction of the SAD work.
//
// For a candidate block B and the anchor A, each split into four 8x8 quadrants,
//     LB(B) = sum_q |sum(A_q) - sum(B_q)|  <=  SAD(A, B)                  (triangle inequality).
// Once some candidate's true SAD is known (UB), every candidate with LB > UB has SAD > UB >= the
// minimum: it can neither win nor tie, so skipping it keeps the reference's "first strict
// minimum in scan order" result bit for bit.
//
// Two kernels run the same phases: k_exh_sea16 (one tile per workgroup) and k_exh_sea16p (persistent
// workgroups that prefetch the next tile into registers, bbme_sea_common.h: persistent_tiles).
// Per tile (NB adjacent macroblocks, one wave each, as in k_exh_qsad16):
//   A  stage the search window of `cur` and the anchors in LDS; quadrant sums of each anchor;
//   A' 8x8 box sums S8 of the staged window, in LDS, two separable passes: the horizontal one is
//      two v_qsad_pk_u16_u8 against a zero reference (four sliding 8-byte sums per lane-op), the
//      vertical one packed u16 adds -- no per-frame table, no extra HBM traffic;
//   B  LB for all NC^2 candidates of the wave's block: four S8 reads, two v_perm_b32 and two
//      v_sad_u16 per candidate; per lane: min LB of each of its R patches (R rows x 4 columns);
//   C  UB := SAD of the candidate with the smallest LB and of the zero vector (64 lanes x 1 dword);
//   D  patches with min LB <= UB go to a workgroup-wide list (LDS atomics);
//   E  the list is processed 64*NB patches at a time, one patch per lane (R x 16 x 4
//      v_qsad_pk_u16_u8 against the anchor read from LDS -- lanes of one wave may serve different
//      blocks), best keys merged with LDS atomicMin; before each chunk entries whose LB exceeds
//      the tightened UB are dropped; template flag E4: four lanes per patch (large windows);
//   F  lane 0 of each wave stores its block's vector.
// On the synthetic and the reference's doc frames 2-6 % of the patches survive.
#include "bbme_sea_common.h"

namespace {

using namespace sea;

#ifdef GME_SEA_STAMPS
#define STAMP(i) do { if (lane == 0) d.stamps[((((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NB + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif


// B: lower bounds of the 4R x R candidates of one lane (candidate rows prow*R + i, columns
// q*4R + 4k + e).  GUARD = false is the branch-free body for blocks whose whole window is inside
// the frame (84 % of them at 720x480, sw = 16); GUARD = true skips candidates outside [lo, hi].
template <int R, bool GUARD>
__device__ __forceinline__ void lower_bounds(const uint64_t* sp0, int XQ, int prow, int q, uint32_t a01, uint32_t a23,
                                             int lo_r, int hi_r, int lo_c, int hi_c, uint32_t (&pkey)[R])
{
#pragma unroll
    for (int k = 0; k < R; ++k) pkey[k] = 0xFFFFFFFFu;
    const uint64_t* top = sp0;
    const uint64_t* bot = sp0 + 8 * XQ;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int ri = prow * R + i;
        if (!GUARD || (ri >= lo_r && ri <= hi_r)) {
            uint64_t t[R + 2], b[R + 2];                   // quads k and k + 2 for k < R: R + 2 distinct ones
#pragma unroll
            for (int k = 0; k < R + 2; ++k) { t[k] = top[k]; b[k] = bot[k]; }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int ci0 = q * 4 * R + 4 * k;
                if (GUARD && (ci0 > hi_c || ci0 + 3 < lo_c)) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (GUARD && (ci0 + e < lo_c || ci0 + e > hi_c)) continue;
                    const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
                    const uint32_t tp = __builtin_amdgcn_perm((uint32_t)(t[k + 2] >> (32 * (e >> 1))), (uint32_t)(t[k] >> (32 * (e >> 1))), sel);
                    const uint32_t bt = __builtin_amdgcn_perm((uint32_t)(b[k + 2] >> (32 * (e >> 1))), (uint32_t)(b[k] >> (32 * (e >> 1))), sel);
                    const uint32_t lb = __builtin_amdgcn_sad_u16(tp, a01, __builtin_amdgcn_sad_u16(bt, a23, 0u));
                    pkey[k] = min(pkey[k], (lb << 13) + (uint32_t)((4 * k + e) * R + i));
                }
            }
        }
        top += XQ;
        bot += XQ;
    }
}

// E, four lanes per patch: lane `sub` of a quad takes anchor rows 4*sub .. 4*sub+3 of the patch's block (R+3 window rows,
// 16*R QSADs); the four partial u16x4 sums are added inside the quad with DPP moves, then lane `sub` turns column `sub`
// of the patch into keys (sad << 13 | scan index).  Returns the quad's smallest key (0xFFFFFFFF: none).  Every lane of
// the wave must call it (DPP); a quad is uniform in `ent` and `active`.
// ent = wave << 25 | lane << 19 | k << 16 | bound: the patch of phase B's lane (prow, q), column group k, of block `wave`.
template <int R>
__device__ __forceinline__ uint32_t eval_patch_quad(const SeaDev& d, const uint32_t* win, const uint32_t* anchor, uint32_t ent,
                                                    bool active, int sub, int trow, int bcol0, int NC)
{
    const int w2 = ent >> 25, l2 = (ent >> 19) & 63, k2 = (ent >> 16) & 7;
    const int wr2 = div_small(w2, d.magic_tc), wc2 = w2 - wr2 * d.tc;     // the patch's block inside the tile
    const int prow2 = l2 >> 2, q2 = l2 & 3;
    uint64_t acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = 0;
    if (active) {
        const uint32_t* lrow = win + (16 * wr2 + prow2 * R + 4 * sub) * d.pitch_dw + wc2 * 4 + q2 * R + k2;
        const uint32_t* an = anchor + w2 * ANCHOR_STRIDE + 16 * sub;
#pragma unroll
        for (int t = 0; t < R + 3; ++t) {
            uint64_t w[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) w[s] = *(const u64_a4*)(lrow + t * d.pitch_dw + s);
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int a = t - i;                   // anchor row 4*sub + a
                if (a < 0 || a > 3) continue;
                const u32x4 ar = *(const u32x4*)(an + a * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i] = __builtin_amdgcn_qsad_pk_u16_u8(w[j], ar[j], acc[i]);
            }
        }
    }
    // sum over the quad: plain 32-bit adds on the packed u16 pairs (a 16x16 SAD is at most 65280, so no partial
    // sum carries into the upper half) -- they take the DPP operand directly, v_pk_add_u16 needs a move first
#pragma unroll
    for (int i = 0; i < R; ++i) {
        uint32_t lo = (uint32_t)acc[i], hi = (uint32_t)(acc[i] >> 32);
        lo += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xF, 0xF, false);            // quad_perm [1,0,3,2]
        hi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xF, 0xF, false);
        lo += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x4E, 0xF, 0xF, false);            // quad_perm [2,3,0,1]
        hi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x4E, 0xF, 0xF, false);
        acc[i] = ((uint64_t)hi << 32) | lo;
    }
    uint32_t key = 0xFFFFFFFFu;
    if (active) {
        // every lane of the quad holds the patch's R x 4 sums now; lane `sub` turns column `sub` into keys
        // (R candidates instead of 4 R on one lane)
        const int c02 = (bcol0 + wc2) * 16, r02 = (trow * d.tr + wr2) * 16;
        const int lo_c = max(0, d.sw - c02), hi_c = min(NC - 1, d.W - 16 - c02 + d.sw);
        const int lo_r2 = max(0, d.sw - r02), hi_r2 = min(NC - 1, d.H - 16 - r02 + d.sw);
        const bool rows_inside2 = NC == 16 * R && lo_r2 == 0 && hi_r2 == NC - 1;
        const int ci = q2 * 4 * R + 4 * k2 + sub, ri0 = prow2 * R;
        const uint32_t shift = (uint32_t)(sub & 1) * 16u;
        if (rows_inside2 && lo_c == 0 && hi_c == NC - 1) {
            // whole window inside the frame: keys relative to the column's first candidate, base added once
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const uint32_t word = (sub & 2) ? (uint32_t)(acc[i] >> 32) : (uint32_t)acc[i];
                const uint32_t sad = __builtin_amdgcn_ubfe(word, shift, 16u);
                key = min(key, (sad << 13) + (uint32_t)i);
            }
            key += (uint32_t)(ci * NC + ri0);
        } else if (ci >= lo_c && ci <= hi_c) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int ri = ri0 + i;
                if (ri < lo_r2 || ri > hi_r2) continue;
                const uint32_t word = (sub & 2) ? (uint32_t)(acc[i] >> 32) : (uint32_t)acc[i];
                const uint32_t sad = __builtin_amdgcn_ubfe(word, shift, 16u);
                key = min(key, (sad << 13) | (uint32_t)(ci * NC + ri));
            }
        }
        // quad minimum (a disabled source lane would hand back `key` itself; quads are uniform in `active`)
        key = min(key, (uint32_t)__builtin_amdgcn_update_dpp((int)key, (int)key, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
        key = min(key, (uint32_t)__builtin_amdgcn_update_dpp((int)key, (int)key, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    }
    return key;
}

// Phases A' .. F of one tile.  On entry the window and the anchors are staged, *count == 0 and the
// workgroup has passed a barrier; there is no barrier after F.
// Returns true (workgroup-uniform) when the tile was handed to the redo kernel instead (SeaDev::redo_list).
template <int R, bool E4>
__device__ __forceinline__ bool tile_phases(const SeaDev& d, uint32_t* lds, const Layout& L, int pair, int trow, int bcol0,
                                            uint32_t mine, uint32_t a01, uint32_t a23, int tid, int tile_id)
{
    const int NB = d.nb, T = blockDim.x;
    const int NC = 2 * d.sw + 16, XQ = d.xq;
    uint32_t* win = lds + L.win;                           // [win_rows][pitch_dw]
    uint32_t* anchor = lds + L.anchor;                     // [NB][ANCHOR_STRIDE]
    uint32_t* best = lds + L.best;                         // [NB] keys, 8 bytes apart (low dword used here)
    uint32_t* count = lds + L.count;
    uint64_t* s8 = (uint64_t*)(lds + L.s8);                // [s8_rows][XQ] packed u16 x 4: S8(y, 4s .. 4s+3)
    uint32_t* work = lds + L.work;                         // [NB*64*R] entries: wave<<25 | lane<<19 | k<<16 | LB
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const WaveBlock wb = wave_block(d, trow, bcol0, wave);
    const int brow = wb.brow, bcol = wb.bcol;
    const bool wave_ok = wb.ok;                            // ragged last tile of a block row / column
    const int r0 = brow * 16, c0 = bcol * 16;
    const int prow = lane >> 2, q = lane & 3;
    (void)NB;

    // ---- A': 8x8 box sums of the window (bbme_sea_common.h) ------------------------------------
#if defined(SEA_ABLATE) && SEA_ABLATE >= 3
    return false;                                          // timing-only build: staging + prep only
#endif
    box_sums8<R>(d, win, s8, tid);
    STAMP(3);
    __syncthreads();
    STAMP(4);
#if defined(SEA_ABLATE) && SEA_ABLATE >= 2
    return false;                                          // timing-only build: + box sums
#endif

    // ---- B: lower bounds of the wave's own block --------------------------------------------
    const int lo_r = max(0, d.sw - r0), hi_r = min(NC - 1, d.H - 16 - r0 + d.sw);
    const bool rows_inside = NC == 16 * R && lo_r == 0 && hi_r == NC - 1;   // and no padding candidates
    uint32_t ub_key = 0xFFFFFFFFu;
    if (wave_ok) {
        const int lo_c = max(0, d.sw - c0), hi_c = min(NC - 1, d.W - 16 - c0 + d.sw);
        // per patch k: min over its candidates of (LB << 13) + local, local = (4k+e)*R + i (any
        // consistent index will do here: the bound only has to name one good candidate)
        uint32_t pkey[R];
        const uint64_t* sp0 = s8 + (16 * wb.wr + prow * R) * XQ + wb.wc * 4 + q * R;
        if (rows_inside && lo_c == 0 && hi_c == NC - 1)                                // wave-uniform
            lower_bounds<R, false>(sp0, XQ, prow, q, a01, a23, lo_r, hi_r, lo_c, hi_c, pkey);
        else
            lower_bounds<R, true>(sp0, XQ, prow, q, a01, a23, lo_r, hi_r, lo_c, hi_c, pkey);
        // pkd[k] = (smallest bound of patch k) << 13 | scan index of the patch's FIRST candidate: no candidate of the
        // patch can have a smaller key (sad << 13 | scan index).  A patch without valid candidates keeps 0xFFFFE000 | index,
        // above every real key (sad <= 65280 < 2^16).
        const uint32_t first_idx = (uint32_t)((q * 4 * R) * NC + prow * R);
        uint32_t pkd[R], lb_key = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            pkd[k] = (pkey[k] & 0xFFFFE000u) | (first_idx + (uint32_t)(4 * k * NC));
            lb_key = min(lb_key, pkey[k]);
        }
        const uint32_t lane_lb = lb_key;                               // this lane's smallest bound << 13 (| local index)
        if (lb_key != 0xFFFFFFFFu) lb_key += (uint32_t)lane << 7;      // local < 4R*R <= 100 < 128
        // ---- C: upper bound from two real candidates (cooperative 16x16 SAD, one dword per lane)
        lb_key = wave_min_u32(lb_key);
        {
            const int bl = (lb_key >> 7) & 63, loc = lb_key & 127;     // decode lane and local index
            const int ce = loc / R, li = loc - ce * R;
            const int idx1 = ((bl & 3) * 4 * R + ce) * NC + (bl >> 2) * R + li;       // min-LB candidate
            const int idx0 = d.sw * NC + d.sw;                                       // zero vector
            const int arow = lane >> 2, aj = lane & 3;
            uint32_t packed = 0;                           // two 16x16 SADs (<= 65280 each) in one register
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const int idx = which ? idx1 : idx0;
                const int ci = idx / NC, ri = idx - ci * NC;
                const int byte = wb.wc * 16 + ci + 4 * aj;
                const uint32_t* p = win + (16 * wb.wr + ri + arow) * d.pitch_dw + (byte >> 2);
                const uint32_t v = __builtin_amdgcn_alignbyte(p[1], p[0], (uint32_t)byte & 3u);
                packed += __builtin_amdgcn_sad_u8(v, mine, 0u) << (16 * which);
            }
            packed = wave_sum_u32(packed);
            ub_key = min(((packed & 0xFFFFu) << 13) | (uint32_t)idx0, ((packed >> 16) << 13) | (uint32_t)idx1);
#ifndef SEA_NO_PROBE_PREV
            // third probe: the vector this wave's block of the PREVIOUS tile of the workgroup ended with (the block one
            // tile to the left, or wherever the schedule came from): spatial coherence makes it a good guess when the
            // smallest bound does not name the best candidate.  Skipped (wave-uniform) when it is one of the two candidates
            // above or lies outside this block's valid window.  Surviving patches 5.4 -> 4.8 % on the synthetic pan, 9.3 ->
            // 8.6 % / 16.8 -> 15.7 % on the real frames; +0.6 .. +1.8 % pairs/s (same-box A/B, round 3).
            const int idxp = (int)(best[2 * wave + 1] & 0x1FFFu);
            const int cip = idxp / NC, rip = idxp - cip * NC;
            if (idxp != idx0 && idxp != idx1 && cip >= lo_c && cip <= hi_c && rip >= lo_r && rip <= hi_r) {
                const int byte = wb.wc * 16 + cip + 4 * aj;
                const uint32_t* p = win + (16 * wb.wr + rip + arow) * d.pitch_dw + (byte >> 2);
                const uint32_t v = __builtin_amdgcn_alignbyte(p[1], p[0], (uint32_t)byte & 3u);
                const uint32_t sadp = wave_sum_u32(__builtin_amdgcn_sad_u8(v, mine, 0u));
                ub_key = min(ub_key, (sadp << 13) | (uint32_t)idxp);
            }
#endif
        }
        // ---- C2 (round 4): ordered evaluation inside a crowded block.  Where the first upper bound leaves more than
        // SeaDev::quota patches (real content: the smallest bound does not name the best candidate, but the best one sits
        // among the small bounds, tools/ub_study.py), the wave scores the up to 16 patches with the smallest bounds itself,
        // four lanes each, before anything is listed: no barrier, a full wave, and phase D then tests the rest against
        // (almost) the block's true minimum.  Any upper bound that is a real candidate's key keeps the result exact
        // (bbme.py:171: only a smaller key replaces the best one).
        uint32_t own_lim = 0;                              // keys below it were scored here (wave-uniform)
        int n_w = 0;                                       // patches the first upper bound leaves (statistics)
#ifndef SEA_NO_C2
        if constexpr (E4) {
            // crowded?  Lanes with a surviving patch are counted first (one ballot); the exact count only where it matters
            if (d.quota > 0 && __popcll(__ballot((lane_lb & 0xFFFFE000u) < ub_key)) > d.engage / 3) {
#pragma unroll
                for (int k = 0; k < R; ++k) n_w += __popcll(__ballot(pkd[k] < ub_key));
                if (n_w > d.engage) {
                    own_lim = first_round_limit(d, lb_key >> 13, ub_key >> 13, 13, [&](uint32_t lim) {
                        int c = 0;
#pragma unroll
                        for (int k = 0; k < R; ++k) c += __popcll(__ballot(pkd[k] < lim));
                        return c;
                    });
                    own_lim = min(own_lim, ub_key);
                    // the chosen patches -> this wave's 16 slots (ties at the smallest bound may exceed them: the surplus
                    // keeps own_rank >= 16 and is listed by phase D like everything else)
                    uint32_t* own = lds + L.own + 16 * wave;
                    int base_rank = 0;
                    uint32_t listed_here = 0;              // bit k: patch k of this lane was scored here
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const bool sel = pkd[k] < own_lim;
                        const unsigned long long m = __ballot(sel);
                        const int rank = base_rank + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (sel && rank < 16) {
                            own[rank] = ((uint32_t)wave << 25) | ((uint32_t)lane << 19) | ((uint32_t)k << 16) | (pkd[k] >> 13);
                            listed_here |= 1u << k;
                        }
                        base_rank += __popcll(m);
                    }
                    const int n_own = min(base_rank, 16);
                    const bool act = (lane >> 2) < n_own;
                    const uint32_t ent = act ? own[lane >> 2] : 0u;
                    uint32_t key = eval_patch_quad<R>(d, win, anchor, ent, act, lane & 3, trow, bcol0, NC);
                    ub_key = min(ub_key, wave_min_u32(key));
#pragma unroll
                    for (int k = 0; k < R; ++k)
                        if (listed_here & (1u << k)) pkd[k] = 0xFFFFFFFFu;       // done: never listed
                    if (lane == 0) atomicAdd(count + 5, (uint32_t)n_own);      // statistics
                }
            }
        }
#endif
        if (lane == 0) best[2 * wave] = ub_key;
        // ---- D: surviving patches -> workgroup list.  A candidate replaces the best one only with a smaller
        // key (sad << 13 | scan index: bbme.py:171 keeps the FIRST strict minimum), and its key is at least
        // LB << 13 | index, so a patch whose smallest possible key -- its bound with the scan index of its first
        // candidate -- does not undercut ub_key holds no winner.  On flat content (every cost ties) this
        // leaves nothing behind the first candidate in scan order instead of everything.
#pragma unroll
        for (int k = 0; k < R; ++k)
            if (pkd[k] < ub_key) {
                const uint32_t slot = atomicAdd(count, 1u);
                work[slot] = ((uint32_t)wave << 25) | ((uint32_t)lane << 19) | ((uint32_t)k << 16) | (pkd[k] >> 13);   // LB <= 65280
            }
        if (own_lim) {                                     // statistics: what the first bound had left and C2 took or pruned
            int pushed = 0;
#pragma unroll
            for (int k = 0; k < R; ++k) pushed += __popcll(__ballot(pkd[k] < ub_key));
            if (lane == 0) atomicAdd(count + 4, (uint32_t)(n_w - pushed));
        }
    }
    STAMP(5);
    __syncthreads();
    STAMP(6);
#ifndef SEA_NO_REDO
    if (d.redo_list && (int)*count > d.redo_threshold) {     // workgroup-uniform: hostile tile, brute force is cheaper
        if (tid == 0) push_redo(d, tile_id, (int)(blockIdx.x & 7));
        return true;
    }
#endif

#if defined(SEA_ABLATE) && SEA_ABLATE >= 1
    return false;                                          // timing-only build: + bounds, UB, list (no evaluation, no result)
#endif
#ifdef SEA_E_CALLS_EVAL
    if constexpr (E4) {
        // ---- E (variant): four lanes per patch (eval_patch_quad) -----------------------------------
        const int n = (int)*count;
        const int sub = lane & 3;
        for (int base = 0; base < n; base += T / 4) {
            const int e = base + (tid >> 2);
            bool active = e < n;
            uint32_t ent = 0;
            if (active) ent = work[e];
            // dropped by a tightened UB (same key rule as phase D); the quad is uniform in `active` (same entry)
            active = active && (((ent & 0xFFFFu) << 13) | (uint32_t)(((int)((ent >> 19) & 3) * 4 * R + 4 * (int)((ent >> 16) & 7)) * NC + (int)((ent >> 21) & 15) * R)) < best[2 * (ent >> 25)];
            const uint32_t key = eval_patch_quad<R>(d, win, anchor, ent, active, sub, trow, bcol0, NC);
            if (active && sub == 0 && key != 0xFFFFFFFFu) atomicMin(&best[2 * (ent >> 25)], key);
            __syncthreads();
        }

#else
    if constexpr (E4) {
        // ---- E (variant): four lanes per patch (the body of eval_patch_quad, kept inline here: the
        // shared function cost this loop 1.2 % in register allocation, same-box A/B round 4) ---------------------------------------------------
        // Lane `sub` of a quad takes anchor rows 4*sub .. 4*sub+3 (R+3 window rows, 16*R QSADs); the four
        // partial u16x4 sums are added inside the quad with DPP moves.
        const int n = (int)*count;
        const int sub = lane & 3;
        for (int base = 0; base < n; base += T / 4) {
            const int e = base + (tid >> 2);
            bool active = e < n;
            uint32_t ent = 0;
            if (active) ent = work[e];
            const int w2 = ent >> 25, l2 = (ent >> 19) & 63, k2 = (ent >> 16) & 7;
            const int wr2 = div_small(w2, d.magic_tc), wc2 = w2 - wr2 * d.tc;     // the patch's block inside the tile
            const int prow2 = l2 >> 2, q2 = l2 & 3;
            // dropped by a tightened UB (same key rule as phase D); the quad is uniform in `active` (same entry),
            // so the DPP exchange below is safe
            active = active && (((ent & 0xFFFFu) << 13) | (uint32_t)((q2 * 4 * R + 4 * k2) * NC + prow2 * R)) < best[2 * w2];
            uint64_t acc[R];
    #pragma unroll
            for (int i = 0; i < R; ++i) acc[i] = 0;
            if (active) {
                const uint32_t* lrow = win + (16 * wr2 + prow2 * R + 4 * sub) * d.pitch_dw + wc2 * 4 + q2 * R + k2;
                const uint32_t* an = anchor + w2 * ANCHOR_STRIDE + 16 * sub;
    #pragma unroll
                for (int t = 0; t < R + 3; ++t) {
                    uint64_t w[4];
    #pragma unroll
                    for (int s = 0; s < 4; ++s) w[s] = *(const u64_a4*)(lrow + t * d.pitch_dw + s);
    #pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const int a = t - i;                   // anchor row 4*sub + a
                        if (a < 0 || a > 3) continue;
                        const u32x4 ar = *(const u32x4*)(an + a * 4);
    #pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i] = __builtin_amdgcn_qsad_pk_u16_u8(w[j], ar[j], acc[i]);
                    }
                }
            }
            // sum over the quad: plain 32-bit adds on the packed u16 pairs (a 16x16 SAD is at most 65280, so no partial
            // sum carries into the upper half) -- they take the DPP operand directly, v_pk_add_u16 needs a move first
    #pragma unroll
            for (int i = 0; i < R; ++i) {
                uint32_t lo = (uint32_t)acc[i], hi = (uint32_t)(acc[i] >> 32);
                lo += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xF, 0xF, false);            // quad_perm [1,0,3,2]
                hi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xF, 0xF, false);
                lo += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x4E, 0xF, 0xF, false);            // quad_perm [2,3,0,1]
                hi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x4E, 0xF, 0xF, false);
                acc[i] = ((uint64_t)hi << 32) | lo;
            }
            if (active) {
                // every lane of the quad holds the patch's R x 4 sums now; lane `sub` turns column `sub` into keys
                // (R candidates instead of 4 R on one lane), the quad's minimum goes to the block's best key
                const int c02 = (bcol0 + wc2) * 16, r02 = (trow * d.tr + wr2) * 16;
                const int lo_c = max(0, d.sw - c02), hi_c = min(NC - 1, d.W - 16 - c02 + d.sw);
                const int lo_r2 = max(0, d.sw - r02), hi_r2 = min(NC - 1, d.H - 16 - r02 + d.sw);
                const bool rows_inside2 = NC == 16 * R && lo_r2 == 0 && hi_r2 == NC - 1;
                const int ci = q2 * 4 * R + 4 * k2 + sub, ri0 = prow2 * R;
                const uint32_t shift = (uint32_t)(sub & 1) * 16u;
                uint32_t key = 0xFFFFFFFFu;
                if (rows_inside2 && lo_c == 0 && hi_c == NC - 1) {
                    // whole window inside the frame: keys relative to the column's first candidate, base added once
    #pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const uint32_t word = (sub & 2) ? (uint32_t)(acc[i] >> 32) : (uint32_t)acc[i];
                        const uint32_t sad = __builtin_amdgcn_ubfe(word, shift, 16u);
                        key = min(key, (sad << 13) + (uint32_t)i);
                    }
                    key += (uint32_t)(ci * NC + ri0);
                } else if (ci >= lo_c && ci <= hi_c) {
    #pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const int ri = ri0 + i;
                        if (ri < lo_r2 || ri > hi_r2) continue;
                        const uint32_t word = (sub & 2) ? (uint32_t)(acc[i] >> 32) : (uint32_t)acc[i];
                        const uint32_t sad = __builtin_amdgcn_ubfe(word, shift, 16u);
                        key = min(key, (sad << 13) | (uint32_t)(ci * NC + ri));
                    }
                }
                // quad minimum (a disabled source lane would hand back `key` itself; quads are uniform in `active`)
                key = min(key, (uint32_t)__builtin_amdgcn_update_dpp((int)key, (int)key, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
                key = min(key, (uint32_t)__builtin_amdgcn_update_dpp((int)key, (int)key, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
                if (sub == 0 && key != 0xFFFFFFFFu) atomicMin(&best[2 * w2], key);
            }
            __syncthreads();
        }

#endif
    } else {
        // ---- E: evaluate the listed patches, one per lane ---------------------------------------
        const int n = (int)*count;
        for (int base = 0; base < n; base += T) {
            const int e = base + tid;
            bool active = e < n;
            uint32_t ent = 0;
            if (active) {
                ent = work[e];
                const uint32_t first = (uint32_t)(((int)((ent >> 19) & 3) * 4 * R + 4 * (int)((ent >> 16) & 7)) * NC + (int)((ent >> 21) & 15) * R);
                active = (((ent & 0xFFFFu) << 13) | first) < best[2 * (ent >> 25)];      // dropped by a tightened UB
            }
            if (active) {
                const int w2 = ent >> 25, l2 = (ent >> 19) & 63, k2 = (ent >> 16) & 7;
                const int wr2 = div_small(w2, d.magic_tc), wc2 = w2 - wr2 * d.tc;     // the patch's block inside the tile
                const int prow2 = l2 >> 2, q2 = l2 & 3;
                const uint32_t* lrow = win + (16 * wr2 + prow2 * R) * d.pitch_dw + wc2 * 4 + q2 * R + k2;
                const uint32_t* an = anchor + w2 * ANCHOR_STRIDE;
                uint64_t acc[R];
    #pragma unroll
                for (int i = 0; i < R; ++i) acc[i] = 0;
    #pragma unroll
                for (int t = 0; t < R + 15; ++t) {
                    uint64_t w[4];
    #pragma unroll
                    for (int s = 0; s < 4; ++s) w[s] = *(const u64_a4*)(lrow + t * d.pitch_dw + s);
    #pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const int a = t - i;
                        if (a < 0 || a > 15) continue;
                        const u32x4 ar = *(const u32x4*)(an + a * 4);
    #pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i] = __builtin_amdgcn_qsad_pk_u16_u8(w[j], ar[j], acc[i]);
                    }
                }
                const int c02 = (bcol0 + wc2) * 16, r02 = (trow * d.tr + wr2) * 16;
                const int lo_c = max(0, d.sw - c02), hi_c = min(NC - 1, d.W - 16 - c02 + d.sw);
                const int lo_r2 = max(0, d.sw - r02), hi_r2 = min(NC - 1, d.H - 16 - r02 + d.sw);
                const bool rows_inside2 = NC == 16 * R && lo_r2 == 0 && hi_r2 == NC - 1;
                const int ci0 = q2 * 4 * R + 4 * k2, ri0 = prow2 * R;
                uint32_t key = 0xFFFFFFFFu;
                if (rows_inside2 && lo_c == 0 && hi_c == NC - 1) {
                    // whole window inside the frame: keys relative to the patch's first candidate, base added once
    #pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4)
    #pragma unroll
                        for (int i = 0; i < R; ++i) {
                            const uint32_t sad = (uint32_t)(acc[i] >> (16 * e4)) & 0xFFFFu;
                            key = min(key, (sad << 13) + (uint32_t)(e4 * NC + i));
                        }
                    key += (uint32_t)(ci0 * NC + ri0);
                } else {
    #pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const int ci = ci0 + e4;
                        if (ci < lo_c || ci > hi_c) continue;
    #pragma unroll
                        for (int i = 0; i < R; ++i) {
                            const int ri = ri0 + i;
                            if (ri < lo_r2 || ri > hi_r2) continue;
                            const uint32_t sad = (uint32_t)(acc[i] >> (16 * e4)) & 0xFFFFu;
                            key = min(key, (sad << 13) | (uint32_t)(ci * NC + ri));
                        }
                    }
                }
                if (key != 0xFFFFFFFFu) atomicMin(&best[2 * w2], key);
            }
            __syncthreads();
        }

    }
    STAMP(7);
    // ---- F: result ------------------------------------------------------------------------------
    if (wave_ok && lane == 0) {
        const int idx = (int)(best[2 * wave] & 0x1FFF);
        const int ci = idx / NC, ri = idx - ci * NC;
        int32_t* o = d.mf + (((long long)pair * d.nbr + brow) * d.nbc + bcol) * 2;
        o[0] = ci - d.sw;
        o[1] = ri - d.sw;
#ifndef SEA_NO_PROBE_PREV
        best[2 * wave + 1] = (uint32_t)idx;                   // next tile's third probe
#endif
    }
    return false;
}

// What the shared persistent driver (bbme_sea_common.h: persistent_tiles) needs from this kernel.
template <int R, bool E4>
struct MaeTile {
    struct Pre { uint32_t a01, a23; };
    static __device__ __forceinline__ Pre prep(const SeaDev&, uint32_t* lds, const Layout& L, int wave, int lane, bool wave_ok, uint32_t mine)
    {
        Pre p = { 0, 0 };
        if (wave_ok) {
            lds[L.anchor + wave * ANCHOR_STRIDE + lane] = mine;
            anchor_quadrants(mine, &p.a01, &p.a23);
        }
        return p;
    }
    static __device__ __forceinline__ bool phases(const SeaDev& d, uint32_t* lds, const Layout& L, int pair, int trow, int bcol0,
                                                  uint32_t mine, const Pre& p, int tid, int tile_id)
    {
        return tile_phases<R, E4>(d, lds, L, pair, trow, bcol0, mine, p.a01, p.a23, tid, tile_id);
    }
};

template <int R, bool E4>
__global__ void __launch_bounds__(1024) k_exh_sea16(SeaDev d)
{
    extern __shared__ uint32_t lds[];
    const Layout L = layout_of(d, R);
    int pair, trow, bcol0;
    if (!locate(d, &pair, &trow, &bcol0)) return;          // whole workgroup
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int NB = d.nb;
    (void)NB;

    STAMP(0);
    // ---- A: window, anchor, quadrant sums ------------------------------------------------
    stage_window(d, lds + L.win, d.cur + (long long)pair * d.plane_stride, bcol0, trow * d.tr * 16);
    uint32_t mine = 0;
    const WaveBlock wb = wave_block(d, trow, bcol0, wave);
    if (wb.ok) {
        const uint8_t* aptr = d.prev + (long long)pair * d.plane_stride + (long long)(wb.brow * 16) * d.pitch + wb.bcol * 16;
        mine = *(const uint32_t*)(aptr + (long long)(lane >> 2) * d.pitch + (lane & 3) * 4);
    }
    const typename MaeTile<R, E4>::Pre pre = MaeTile<R, E4>::prep(d, lds, L, wave, lane, wb.ok, mine);
    if (lane == 0) lds[L.best + 2 * wave + 1] = (uint32_t)(d.sw * (2 * d.sw + 16) + d.sw);      // no previous tile: the zero vector
    if (threadIdx.x == 0) { lds[L.count] = 0; lds[L.count + 4] = 0; lds[L.count + 5] = 0; }
    STAMP(1);
    __syncthreads();
    STAMP(2);
    MaeTile<R, E4>::phases(d, lds, L, pair, trow, bcol0, mine, pre, (int)threadIdx.x, tile_number(d, pair, trow, bcol0));
    if (threadIdx.x == 0) {                                // the counts are final behind phase D's barrier
        atomicAdd(d.status + GME_STATUS_STATS + 16 * (blockIdx.x & 7), lds[L.count] + lds[L.count + 5]);
        atomicAdd(d.status + GME_STATUS_STATS + 16 * (blockIdx.x & 7) + 2, lds[L.count] + lds[L.count + 4]);
    }
}

template <int R, int NV, int GEO = 0>
__global__ void __launch_bounds__(1024, (R <= 3 ? 8 : 6)) k_exh_sea16p(SeaDev d)
{
    extern __shared__ uint32_t lds[];
    fix_geometry<R, GEO>(d);
    const Layout L = layout_of(d, R);
    if ((threadIdx.x & 63) == 0) lds[L.best + 2 * (threadIdx.x >> 6) + 1] = (uint32_t)(d.sw * (2 * d.sw + 16) + d.sw);   // third probe of the first tile: the zero vector
    persistent_tiles<NV, MaeTile<R, (R >= 3)>>(d, lds, L);
}

}  // namespace

bool bbme_sea_applies(int bs, int sw, int procedure, int pnorm)
{
    if (procedure != GME_SEARCH_EXHAUSTIVE || bs != 16 || pnorm != GME_NORM_MAE) return false;
    if (sw < 0 || sw % 4 != 0) return false;
    const int NC = 2 * sw + 16;
    return (NC + 15) / 16 <= 5 && NC * NC <= 8192 && !getenv("GME_FORCE_GENERIC") && !getenv("GME_EXH_BRUTE");
}

#ifdef GME_SEA_STAMPS
static long long* g_stamps = nullptr;
#endif

int launch_bbme_sea(gme_ctx* ctx, const BbmeJob& job, bool* handled)
{
    *handled = false;
    if (!bbme_sea_applies(job.bs, job.sw, job.procedure, job.pnorm)) return GME_OK;
    const int NC = 2 * job.sw + 16, R = (NC + 15) / 16;
    const int nbr = job.H / 16, nbc = job.W / 16;
    if (nbr == 0 || nbc == 0) return GME_OK;
    SeaDev d;
    d.status = (uint32_t*)ctx->status; d.dynamic = 0;
    d.prev = job.prev; d.cur = job.cur; d.plane_stride = job.plane_stride;
    d.pairs = job.pairs; d.H = job.H; d.W = job.W; d.pitch = job.pitch; d.sw = job.sw;
    d.nbr = nbr; d.nbc = nbc; d.mf = job.mf;
#ifdef GME_SEA_STAMPS
    d.stamps = g_stamps;
#endif
    size_t lds = 0;
    GME_REQUIRE(plan(R, nbr, nbc, job.sw, &d, &lds), GME_ERR_ARG, "search window too large for LDS");
    d.sqbox = nullptr; d.sqbox_stride = 0;
    const dim3 block(64 * d.nb);
    // hostile tiles (bound prunes little) -> brute-force redo kernel behind this one; GME_SEA_REDO=0 switches it off,
    // GME_SEA_REDO_FRAC sets the share of a tile's patches from which phase E costs more than evaluating everything
    d.redo_list = nullptr; d.redo_threshold = 0x7FFFFFFF;
    // ordered evaluation inside crowded blocks (phase C2; bbme_sea_common.h: SeaDev::quota); GME_SEA_QUOTA=0 switches it off
    d.quota = getenv("GME_SEA_QUOTA") ? atoi(getenv("GME_SEA_QUOTA")) : SEA_DEFAULT_QUOTA;
    d.bisect = getenv("GME_SEA_BISECT") ? atoi(getenv("GME_SEA_BISECT")) : SEA_DEFAULT_BISECT;
    d.engage = getenv("GME_SEA_ENGAGE") ? atoi(getenv("GME_SEA_ENGAGE")) : SEA_DEFAULT_ENGAGE;
    if (d.engage < d.quota) d.engage = d.quota;
    const bool redo = !(getenv("GME_SEA_REDO") && atoi(getenv("GME_SEA_REDO")) == 0);
    if (redo) {
        int rc = ctx_redo_list(ctx, (size_t)job.pairs * d.wg_per_pair, &d.redo_list);
        if (rc) return rc;
        const double frac = getenv("GME_SEA_REDO_FRAC") ? atof(getenv("GME_SEA_REDO_FRAC")) : REDO_DEFAULT_FRAC;
        d.redo_threshold = (int)(frac * d.nb * 64 * R);
        if (!job.status_fresh) GME_HIP_TRY(hipMemsetAsync(d.status + GME_STATUS_REDO, 0, 2 * sizeof(uint32_t), ctx->stream));
    }
    const PersistPlan pp = plan_persistent(d, lds, job.pairs, ctx->prop.multiProcessorCount);
    const int nv = pp.nv;
    // R = 2, 3 with more than 8 staging rows per thread would spill the prefetched tile (64 VGPRs at
    // 8 waves/SIMD); those shapes keep the one-tile kernel
    const bool fits = R <= 1 || R >= 4 || nv <= 8;
    if (pp.use && fits) {
        const dim3 grid((unsigned)(8 * pp.g));
        if (pp.dynamic) {
            d.dynamic = 1;
            if (!job.status_fresh) GME_HIP_TRY(hipMemsetAsync(d.status + GME_STATUS_TILECTR, 0, 8 * 16 * sizeof(uint32_t), ctx->stream));
        }
        // the two BASELINE shapes (720x480 sw 16: 2x4 tiles; 1080p sw 32: 2x6 tiles) have instances with the tile
        // geometry folded in at compile time; GME_SEA_GENERIC=1 keeps the run-time form (A/B, tests)
        const bool fixed_ok = !getenv("GME_SEA_GENERIC");
        const bool fix3 = fixed_ok && R == 3 && nv <= 5 && geometry_matches(d, 3, 2 * 16 + 4);
        const bool fix5 = fixed_ok && R == 5 && nv <= 7 && geometry_matches(d, 5, 2 * 16 + 6);
        plan_note(ctx, (long long)job.pairs * nbr * nbc * 64 * R, "k_exh_sea16p<%d,%d> tiles %dx%d persistent-%s%s grid %u lds %zu",
                  R, fix3 ? 5 : fix5 ? 7 : nv <= 6 ? 6 : nv <= 8 ? 8 : nv <= 12 ? 12 : 16, d.tr, d.tc, pp.dynamic ? "dynamic" : "static", (fix3 || fix5) ? " geometry-fixed" : "", grid.x, lds);
        if (fix3) {
            hipLaunchKernelGGL((k_exh_sea16p<3, 5, 2 * 16 + 4>), grid, block, lds, ctx->stream, d);
        } else if (fix5) {
            hipLaunchKernelGGL((k_exh_sea16p<5, 7, 2 * 16 + 6>), grid, block, lds, ctx->stream, d);
        } else
#define SEA_LAUNCH_P(RR, NVV) hipLaunchKernelGGL((k_exh_sea16p<RR, NVV>), grid, block, lds, ctx->stream, d)
#define SEA_LAUNCH_PN(RR) do { if (nv <= 6) SEA_LAUNCH_P(RR, 6); else if (nv <= 8) SEA_LAUNCH_P(RR, 8); \
                               else if (nv <= 12) SEA_LAUNCH_P(RR, 12); else SEA_LAUNCH_P(RR, 16); } while (0)
        switch (R) {
        case 1: SEA_LAUNCH_PN(1); break;
        case 2: SEA_LAUNCH_PN(2); break;
        case 3: SEA_LAUNCH_PN(3); break;
        case 4: SEA_LAUNCH_PN(4); break;
        default: SEA_LAUNCH_PN(5); break;
        }
#undef SEA_LAUNCH_PN
#undef SEA_LAUNCH_P
    } else {
    dim3 grid;
    GME_REQUIRE(grid_for(d, &grid), GME_ERR_ARG, "too many workgroups in one launch");
    plan_note(ctx, (long long)job.pairs * nbr * nbc * 64 * R, "k_exh_sea16<%d> tiles %dx%d one-tile grid %ux%ux%u lds %zu", R, d.tr, d.tc,
              grid.x, grid.y, grid.z, lds);
    // phase E with four lanes per patch (a quarter of the latency the other waves wait for): +11 % at
    // sw 32, +3 % at sw 16 in the persistent kernel; GME_SEA_E4 = 0 / 1 overrides it for this one-tile kernel
    const bool e4 = getenv("GME_SEA_E4") ? atoi(getenv("GME_SEA_E4")) != 0 : R >= 3;
#define SEA_LAUNCH(RR) do { if (e4) hipLaunchKernelGGL((k_exh_sea16<RR, true>), grid, block, lds, ctx->stream, d); \
                            else hipLaunchKernelGGL((k_exh_sea16<RR, false>), grid, block, lds, ctx->stream, d); } while (0)
    switch (R) {
    case 1: SEA_LAUNCH(1); break;
    case 2: SEA_LAUNCH(2); break;
    case 3: SEA_LAUNCH(3); break;
    case 4: SEA_LAUNCH(4); break;
    default: SEA_LAUNCH(5); break;
    }
#undef SEA_LAUNCH
    }
    GME_HIP_TRY(hipGetLastError());
    *handled = true;
    if (redo) return launch_exh_redo(ctx, job, R, d.tr, d.tc, d.wg_per_row, d.wg_per_pair, d.redo_list, d.status + GME_STATUS_REDO, d.status + GME_STATUS_REDO + 1);
    return GME_OK;
}
