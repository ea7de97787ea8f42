// Diagnostic build of k_exh_sea16 with s_memtime stamps at its phase boundaries: prints the mean
// cycles a wave spends in each phase (and waiting at each barrier) for a 720x480 batch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGME_SEA_STAMPS \
//         -I global-motion-estimation_amd/csrc tools/microbench/sea_phases.hip -o tools/microbench/sea_phases
// Never quote this build's run time: the stamps change scheduling (guide §7, In-kernel stamps).
#include <stdarg.h>
#include <vector>
#include "bbme_sea.hip"

void gme_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int H = 480, W = 720, pitch = 768, pairs = argc > 1 ? atoi(argv[1]) : 256, sw = 16;
    const long long stride = (long long)pitch * H;
    std::vector<uint8_t> host((size_t)stride * (pairs + 1));
    // smooth-ish random texture moving by (5,-3) per frame, like the bench sequence
    std::vector<uint8_t> canvas((size_t)1024 * 2048);
    uint32_t rs = 12345;
    for (auto& v : canvas) { rs = rs * 1664525u + 1013904223u; v = (uint8_t)(rs >> 24); }
    for (int f = 0; f <= pairs; ++f)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const int cy = (y + 3 * f) & 1023, cx = (x - 5 * f) & 2047;
                int s = 0;
                for (int dy = 0; dy < 3; ++dy) for (int dx = 0; dx < 3; ++dx) s += canvas[(size_t)((cy + dy) & 1023) * 2048 + ((cx + dx) & 2047)];
                host[(size_t)f * stride + (size_t)y * pitch + x] = (uint8_t)(s / 9);
            }
    uint8_t* dev; int32_t* mf; long long* stamps;
    CK(hipMalloc(&dev, host.size() + pitch)); CK(hipMemcpy(dev, host.data(), host.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&mf, (size_t)pairs * 30 * 45 * 2 * 4));
    gme_ctx ctx; CK(hipStreamCreate(&ctx.stream)); CK(hipGetDeviceProperties(&ctx.prop, 0));
    setenv("GME_SEA_PERSIST", "0", 1);                 // the stamps index by the plain kernel's 3-D grid
    BbmeJob job; job.prev = dev; job.cur = dev + stride; job.plane_stride = stride; job.pairs = pairs; job.H = H; job.W = W;
    job.pitch = pitch; job.bs = 16; job.sw = sw; job.procedure = 0; job.pnorm = 0; job.mf = mf; job.sqbox_cur = nullptr; job.sqbox_stride = 0;
    // replicate launch_bbme_sea's setup, plus the stamp buffer
    const size_t nstamp = (size_t)((pairs + 7) / 8) * 8 * 30 * 3 * 16 * 8;
    CK(hipMalloc(&stamps, nstamp * 8)); CK(hipMemset(stamps, 0, nstamp * 8));
    g_stamps = stamps;
    bool handled = false;
    for (int rep = 0; rep < 3; ++rep) if (launch_bbme_sea(&ctx, job, &handled) != 0 || !handled) { printf("launch failed\n"); return 1; }
    CK(hipStreamSynchronize(ctx.stream));
    std::vector<long long> st(nstamp);
    CK(hipMemcpy(st.data(), stamps, nstamp * 8, hipMemcpyDeviceToHost));
    double sum[7] = { 0 }; long long n = 0;
    for (size_t w = 0; w < nstamp / 8; ++w) {
        if (st[w * 8 + 7] == 0 || st[w * 8] == 0) continue;
        for (int i = 0; i < 7; ++i) sum[i] += (double)(st[w * 8 + i + 1] - st[w * 8 + i]);
        ++n;
    }
    const char* names[7] = { "A stage+anchor", "barrier 1", "A' box sums", "barrier 2", "B+C+D bounds/UB/list", "barrier 3", "E patches (+barriers)" };
    double tot = 0; for (int i = 0; i < 7; ++i) tot += sum[i] / n;
    printf("waves %lld, mean cycles per wave: total %.0f\n", n, tot);
    for (int i = 0; i < 7; ++i) printf("  %-26s %8.0f  %5.1f%%\n", names[i], sum[i] / n, 100 * sum[i] / n / tot);
    return 0;
}
