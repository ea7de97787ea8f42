// Internal declarations shared by the translation units of libgme_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gme_hip.h"

void gme_set_error(const char* fmt, ...);

#define GME_HIP_TRY(expr)                                                                  \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            gme_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                          __LINE__);                                                       \
            return GME_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define GME_REQUIRE(cond, code, ...)       \
    do {                                   \
        if (!(cond)) {                     \
            gme_set_error(__VA_ARGS__);    \
            return (code);                 \
        }                                  \
    } while (0)

// An image plane (or a stack of equally sized planes) in HBM.
// pitch is a multiple of 64 bytes; bytes between W and pitch are zero.
struct Plane {
    uint8_t* ptr = nullptr;
    int H = 0, W = 0, pitch = 0;
    int64_t stride = 0;   // bytes between consecutive planes of a stack
    int count = 0;
    size_t bytes() const { return (size_t)stride * (size_t)count; }
    uint8_t* at(int i) const { return ptr + (int64_t)i * stride; }
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct gme_ctx {
    std::mutex mu;                // held by every C-ABI entry point for the whole call (gme_api.hip: GME_ENTER)
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipDeviceProp_t prop;
    // growable device scratch for the single-pair convenience calls
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    int* status = nullptr;        // device words (GME_STATUS_WORDS): [0] set by kernels whose safety guards trip;
                                  // [GME_STATUS_TILECTR ..] per-XCD tile counters of the persistent search kernels,
                                  // [GME_STATUS_STATS ..] per-XCD statistics of the last block-matching call (16 words apart)
    void* comm = nullptr;         // RCCL communicator (gme_comm.hip), one per context = per rank
    int comm_rank = 0, comm_world = 0;
    hipStream_t copy_stream = nullptr, copy_stream2 = nullptr, back_stream = nullptr;   // gme_seq_bbme_streamed: uploads / read-backs beside the kernels
    uint8_t* stage = nullptr;        // device staging of tight host frames (gme_seq_bbme_streamed), repacked into the planes
    size_t stage_bytes = 0;
    uint32_t* redo_list = nullptr;   // tiles the elimination kernels hand to the brute-force redo kernel (grown on demand)
    size_t redo_cap = 0;             // entries
    char plan[192] = "";          // kernel / tile shape / schedule the last block-matching call chose (gme_last_bbme_info)
    long long plan_patches = 0;   // candidate patches that call's bound was applied to (0: a kernel without elimination)
};
constexpr int GME_STATUS_WORDS = 1024, GME_STATUS_TILECTR = 64, GME_STATUS_STATS = 256;
constexpr int GME_STATUS_REDO = 512;           // [0] tiles listed for the redo kernel, [1] items it has drawn
int ctx_redo_list(gme_ctx* ctx, size_t entries, uint32_t** out);

void plan_note(gme_ctx* ctx, long long patches, const char* fmt, ...);

int ctx_scratch(gme_ctx* ctx, size_t bytes, void** out);
int plane_alloc(gme_ctx* ctx, Plane* p, int count, int H, int W);
void plane_free(Plane* p);

struct FitLevelBuf {
    int h = 0, w = 0;             // motion-field shape at this level
    int32_t* gt = nullptr;        // [P][h][w][2]
    int16_t* model = nullptr;     // [P][h][w][2]
    uint8_t* mask = nullptr;      // [P][h][w]
    int32_t* diff = nullptr;      // [P][h][w] L1 distance gt vs model
    void* list = nullptr;         // [P][h][w] int4 inlier list, only for fields too large for LDS
    int32_t* thr = nullptr;       // [P]
    double* sums = nullptr;       // [P][15]
};

struct gme_seq {
    gme_ctx* ctx = nullptr;
    int N = 0, H = 0, W = 0;      // N: frames in use (gme_seq_set_frames), <= N_cap
    int N_cap = 0;                // frames the sequence was created for: every buffer is sized for it
    Plane level[3];               // level[2] = full resolution, [1], [0] = pyramid
    bool pyramids_valid = false;
    // generic BBME result
    int32_t* mv = nullptr;        // [P][h][w][2]
    size_t mv_bytes = 0;
    int mv_h = 0, mv_w = 0, mv_pairs = 0;
    uint32_t* sqbox[3] = { nullptr, nullptr, nullptr };   // per level: 16x16 box sums of squares per frame (MSE fast path)
    size_t sqbox_bytes[3] = { 0, 0, 0 };
    bool sqbox_valid[3] = { false, false, false };
    int sqbox_kind[3] = { 0, 0, 0 };
    // GME state
    int gme_fd = 0, gme_bs = 0, gme_pairs = 0;
    int gme_procedure = 0, gme_sw = 0;
    bool bbme_pending[3] = { false, false, false };   // level searches gme_seq_gme_begin deferred (see there)
    bool split_phase = false;     // gme_seq_set_split_phase: begin / fit / compensate return once their work is queued
    hipEvent_t ready = nullptr;   // recorded behind the last result copy of such a call; gme_seq_wait waits on it
    hipEvent_t upload_gate = nullptr, uploaded = nullptr;   // split-phase gme_seq_upload: ties the shared upload stream to this context's stream
    FitLevelBuf fit[3];           // fit[0].gt = dense field
    FitLevelBuf fit_mv;           // stage buffers for fitting `mv` directly (gt not owned)
    int fit_mv_pairs = 0;
    double* mv_params = nullptr;  // [P][6] parameters for fit_mv
    size_t mv_params_bytes = 0;
    float* params0 = nullptr;     // [P][6]
    double* params_in = nullptr;  // [P][6]
    int32_t* solve_flags = nullptr;      // [P] gme_seq_gme_device_solve: pairs whose device solve must be redone on the host
    size_t gme_alloc_pairs = 0;
    // compensation
    Plane comp;                   // [P] compensated frames
    double* comp_params = nullptr;       // [P][6]
    unsigned long long* sse = nullptr;   // [P]
    // per-pair summary rows of `mv` (gme_seq_mv_summary) and their all-gather over the ranks (gme_seq_mv_summary_gather)
    double* summary = nullptr;           // [n_max][6], zero-padded behind mv_pairs rows
    size_t summary_bytes = 0;
    double* gathered = nullptr;          // [world][n_max][6]
    size_t gathered_bytes = 0;
    uint8_t* synth_canvas = nullptr;
    uint64_t synth_seed = 0;
    bool synth_valid = false;
};

// ---- kernel launchers (bbme_kernels.hip) -----------------------------------
struct BbmeJob {
    const uint8_t* prev;          // first "previous" plane
    const uint8_t* cur;           // first "current" plane
    int64_t plane_stride;         // bytes between consecutive pairs' planes (same for prev/cur)
    int pairs;
    int H, W, pitch;
    int bs, sw, procedure, pnorm;
    int32_t* mf;                  // [pairs][H/bs][W/bs][2]
    const uint32_t* sqbox_cur;    // optional, matches `cur` planes: [pairs][H][pitch] uint32 (SqTable)
    int64_t sqbox_stride;         // elements between consecutive planes
    bool chained = false;         // a later chunk of one streamed call: keep the plan text and the statistics
    bool status_fresh = false;    // launch_bbme has just cleared the tile counters and redo words (first chunk of a call)
};
int launch_bbme(gme_ctx* ctx, const BbmeJob& job);
int launch_exh_redo(gme_ctx* ctx, const BbmeJob& job, int R, int tr, int tc, int tile_wg_per_row, int tile_wg_per_pair,
                    const uint32_t* list, const uint32_t* count, uint32_t* head);
// The table of 16x16 box sums of squares (exhaustive MSE, bs 16): uint32 [H][pitch] per frame, positions are those of the
// frame (row * pitch + column); rows > H - 16 and columns > W - 16 hold nothing.  The four positions x .. x+3 of a row
// (x % 4 == 0) come with one 16-byte read.  (A 24-bit layout -- 16-bit and 8-bit planes, 3 instead of 4 bytes per position --
// was tried in round 3: the table kernel's two narrower stores per row made IT 17 % slower, 1.43 against 1.22 ms per 2049
// frames of 720x480, and the whole MSE search 4 %.)
#ifdef __HIPCC__
// Wave-wide reductions without LDS traffic: four DPP steps leave every lane with the result of its 16-lane row (xor 1,
// xor 2 inside quads, then the two mirrors); row_bcast:15 folds rows 0 and 2 into rows 1 and 3, row_bcast:31 folds row 1
// into row 3, and one v_readlane of lane 63 makes the result scalar (round 4: 7 instead of 11 instructions; rounds 1-3
// read one lane of every row and combined them on the scalar side).
#define GME_DPP(v, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xF, 0xF, false))
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    v = min(v, GME_DPP(v, 0xB1));                          // quad_perm [1,0,3,2]
    v = min(v, GME_DPP(v, 0x4E));                          // quad_perm [2,3,0,1]
    v = min(v, GME_DPP(v, 0x141));                         // row_half_mirror
    v = min(v, GME_DPP(v, 0x140));                         // row_mirror
#ifdef GME_WAVE_REDUCE_READLANES
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(r0, r1), min(r2, r3));
#else
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xA, 0xF, false));     // row_bcast:15 -> rows 1, 3
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xC, 0xF, false));     // row_bcast:31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
#endif
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    v += GME_DPP(v, 0xB1);
    v += GME_DPP(v, 0x4E);
    v += GME_DPP(v, 0x141);
    v += GME_DPP(v, 0x140);
#ifdef GME_WAVE_REDUCE_READLANES
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
#else
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);             // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);             // row_bcast:31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
#endif
}

typedef const uint32_t* SqTable;
__device__ __forceinline__ SqTable sq_table(const uint32_t* slot, int, int) { return slot; }
__device__ __forceinline__ uint32_t sq1(SqTable t, long long idx) { return t[idx]; }
__device__ __forceinline__ void sq4(SqTable t, long long idx, uint32_t (&out)[4])       // idx % 4 == 0
{
    const uint4 v = *(const uint4*)(t + idx);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
#endif
int launch_sqbox16(gme_ctx* ctx, const uint8_t* src, long long src_stride, int count, int H, int W, int pitch,
                   uint32_t* out, long long stride, bool sgn);
// per-frame auxiliary table a fast exhaustive kernel wants for `cur` (BbmeJob::sqbox_cur):
// 0 none, 1 = 16x16 box sums of squares (MSE, k_exh_dot16 and the elimination kernels), 2 = 16x16 box sums of the squares
// of (byte - 128) (MSE on the matrix cores, bbme_mfma.hip)
int bbme_aux_kind(int bs, int sw, int procedure, int pnorm);
int launch_aux_table(gme_ctx* ctx, int kind, const uint8_t* src, long long src_stride, int count, int H, int W,
                     int pitch, uint32_t* out, long long stride);
bool bbme_sea_applies(int bs, int sw, int procedure, int pnorm);
// ---- bbme_mfma.hip: exhaustive MSE at bs 16 as an int8 correlation on the matrix cores
bool bbme_mfma_wanted(int sw);                // search windows it takes (and GME_EXH_MFMA has not switched it off)
int launch_bbme_mfma(gme_ctx* ctx, const BbmeJob& job, bool* handled);

int bbme_check_args(int H, int W, int bs, int sw, int procedure, int pnorm);

// ---- gme_kernels.hip --------------------------------------------------------
int max_grid_planes();
int launch_repack(gme_ctx* ctx, hipStream_t stream, const uint8_t* src, int count, int H, int W, uint8_t* dst, int pitch,
                  long long dst_stride);
int launch_pyrdown(gme_ctx* ctx, const Plane& src, const Plane& dst);
int launch_first_params(gme_ctx* ctx, const int32_t* dense, int pairs, int n_blocks, float* params0);
int launch_project_first(gme_ctx* ctx, const float* params0, int pairs, double* params_in);
int launch_fit_level(gme_ctx* ctx, const int32_t* gt, int pairs, int h, int w, const double* params,
                     int drop_count, int level_H, int level_W, int16_t* model, uint8_t* mask,
                     int32_t* diff, int32_t* thr, double* sums, void* list);
int launch_affine_field(gme_ctx* ctx, const double* params, int pairs, int h, int w, int16_t* out);
int launch_solve3(gme_ctx* ctx, const double* sums, int pairs, int project, int h, int w, double* params_out, int32_t* flags, int flag_bit);
int launch_mv_summary(gme_ctx* ctx, const int32_t* mf, int pairs, int n_blocks, double* rows);
int launch_compensate(gme_ctx* ctx, const uint8_t* frames, int64_t frame_stride, int pairs, int H,
                      int W, int pitch, const int32_t* mf32, const double* params, int h, int w,
                      uint8_t* out, int64_t out_stride, int out_pitch, const uint8_t* cur,
                      int64_t cur_stride, unsigned long long* sse);
int launch_sse(gme_ctx* ctx, const uint8_t* a, int64_t a_stride, int a_pitch, const uint8_t* b,
               int64_t b_stride, int b_pitch, int pairs, int H, int W, unsigned long long* sse);

// ---- synth_kernels.hip ------------------------------------------------------
int launch_synth_canvas(gme_ctx* ctx, uint64_t seed, uint8_t* canvas);
int launch_synth_frames(gme_ctx* ctx, uint64_t seed, int t0, const uint8_t* canvas, const Plane& dst);
