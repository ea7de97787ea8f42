// On-device form of the deterministic synthetic sequence of SURVEY.md §8(d)
// (host definition: global-motion-estimation_amd/synth.py; both must agree bit for bit).
#include "gme_internal.h"

namespace {

constexpr int CANVAS_H = 2048, CANVAS_W = 4096;

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ unsigned long long hash64(unsigned long long seed, unsigned long long k)
{
    return splitmix64(seed * 0x9E3779B97F4A7C15ull + k);
}

__global__ void __launch_bounds__(256) k_canvas(unsigned long long seed, uint8_t* canvas)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    int box = 0;
    for (int dy = -2; dy <= 2; ++dy) {
        const int yy = (y + dy) & (CANVAS_H - 1);
        for (int dx = -2; dx <= 2; ++dx) {
            const int xx = (x + dx) & (CANVAS_W - 1);
            box += (int)(hash64(seed, (unsigned long long)yy * CANVAS_W + xx) & 0xFF);
        }
    }
    box /= 25;
    const int cell = (int)(hash64(seed + 1, (unsigned long long)(y / 32) * 128 + (x / 32)) & 0xFF);
    canvas[(long long)y * CANVAS_W + x] = (uint8_t)((box + cell) / 2);
}

__device__ __forceinline__ int pmod(long long a, int m)
{
    long long r = a % m;
    return (int)(r < 0 ? r + m : r);
}

__global__ void __launch_bounds__(256) k_frames(unsigned long long seed, int t0, const uint8_t* canvas, uint8_t* dst,
                                                long long stride, int H, int W, int pitch)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const long long t = (long long)t0 + blockIdx.z;
    int v = canvas[(long long)pmod(y + 3 * t, CANVAS_H) * CANVAS_W + pmod(x - 5 * t, CANVAS_W)];
    const int rh = H / 4, rw = W / 6;
    if (rh > 0 && rw > 0) {
        // rectangle rows (H/3 + 4t + yy) mod H, yy < rh: invert for this pixel
        const int yy = pmod((long long)y - (H / 3 + 4 * t), H);
        const int xx = pmod((long long)x - (W / 3 - 7 * t), W);
        if (yy < rh && xx < rw) v = (int)(hash64(seed + 2, (unsigned long long)yy * rw + xx) & 0xFF);
    }
    const int noise = (int)(hash64(seed + 3 + (unsigned long long)t, (unsigned long long)y * W + x) % 5) - 2;
    v += noise;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    dst[(long long)blockIdx.z * stride + (long long)y * pitch + x] = (uint8_t)v;
}

}  // namespace

int launch_synth_canvas(gme_ctx* ctx, uint64_t seed, uint8_t* canvas)
{
    hipLaunchKernelGGL(k_canvas, dim3(CANVAS_W / 256, CANVAS_H), dim3(256), 0, ctx->stream,
                       (unsigned long long)seed, canvas);
    GME_HIP_TRY(hipGetLastError());
    return GME_OK;
}

int launch_synth_frames(gme_ctx* ctx, uint64_t seed, int t0, const uint8_t* canvas, const Plane& dst)
{
    for (int first = 0; first < dst.count; first += 32768) {
        const int cnt = dst.count - first < 32768 ? dst.count - first : 32768;
        const dim3 grid((dst.W + 63) / 64, (dst.H + 3) / 4, cnt);
        hipLaunchKernelGGL(k_frames, grid, dim3(256), 0, ctx->stream, (unsigned long long)seed, t0 + first, canvas,
                           dst.at(first), (long long)dst.stride, dst.H, dst.W, dst.pitch);
        GME_HIP_TRY(hipGetLastError());
    }
    return GME_OK;
}
