// Microbenchmark: issue rate of the integer VALU ops the block-matching kernels are
// built from (v_sad_u8, v_qsad_pk_u16_u8, v_mqsad_u32_u8, v_dot4_u32_u8, v_alignbyte_b32)
// on gfx950, plus a semantic check of v_qsad_pk_u16_u8 against a host model.
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 2048;
constexpr int NACC = 8;

template <int OP, bool SGPR>
__global__ void rate_kernel(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, long long* cycles, uint32_t suni)
{
    uint32_t x = in[threadIdx.x & 63];
    uint32_t y = SGPR ? suni : in[64 + (threadIdx.x & 63)];
    uint64_t w = ((uint64_t)in[128 + (threadIdx.x & 63)] << 32) | x;
    uint32_t a32[NACC];
    uint64_t a64[NACC];
    for (int i = 0; i < NACC; ++i) { a32[i] = i; a64[i] = i; }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (OP == 0) a32[i] = __builtin_amdgcn_sad_u8(x, y, a32[i]);
            if (OP == 1) a64[i] = __builtin_amdgcn_qsad_pk_u16_u8(w, y, a64[i]);
            if (OP == 2) a64[i] = __builtin_amdgcn_mqsad_pk_u16_u8(w, y, a64[i]);
            if (OP == 3) a32[i] = __builtin_amdgcn_udot4(x, y, a32[i], false);
            if (OP == 4) a32[i] = __builtin_amdgcn_alignbyte(a32[i], x, y);
            if (OP == 5) a32[i] = a32[i] + x;
            if (OP == 6) a32[i] = __builtin_amdgcn_sad_u16(x, y, a32[i]);
            if (OP == 7) a32[i] = __builtin_amdgcn_msad_u8(x, y, a32[i]);
        }
        // keep x changing so nothing is hoisted, 1 extra VALU op per NACC
        x += 0x01010101u;
        w += 0x0101010101010101ull;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    uint64_t s = 0;
    for (int i = 0; i < NACC; ++i) s += a32[i] + a64[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP, bool SGPR>
void run(const char* name, const uint32_t* din, uint64_t* dout, long long* dcyc, int waves_per_simd)
{
    int threads = 64 * 4 * waves_per_simd > 1024 ? 1024 : 64 * 4 * waves_per_simd;   // waves per WG
    int wg_per_cu = (64 * 4 * waves_per_simd) / threads;
    int blocks = 256 * wg_per_cu;
    int nwaves = blocks * threads / 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<OP, SGPR><<<blocks, threads>>>(din, dout, dcyc, 0x12345678u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) rate_kernel<OP, SGPR><<<blocks, threads>>>(din, dout, dcyc, 0x12345678u);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    std::vector<long long> cyc(nwaves);
    CK(hipMemcpy(cyc.data(), dcyc, sizeof(long long) * nwaves, hipMemcpyDeviceToHost));
    double avg = 0; for (auto c : cyc) avg += c; avg /= nwaves;
    double instr = (double)ITERS * NACC;
    // s_memtime ticks at a fixed 100 MHz on CDNA? report both raw ticks and wall-derived numbers
    double wave_instr_per_s = instr * nwaves / (ms * 1e-3);
    printf("%-22s sgpr=%d waves/simd=%d  %.3f ms  ticks/wave=%.0f  wave-instr/s=%.3e  lane-ops/s=%.3e  ns per instr per SIMD=%.3f\n",
           name, (int)SGPR, waves_per_simd, ms, avg, wave_instr_per_s, wave_instr_per_s * 64, 1e9 / (wave_instr_per_s / 1024));
}

__global__ void sem_kernel(const uint32_t* in, uint64_t* out, uint32_t* out32)
{
    int t = threadIdx.x;
    uint64_t w = ((uint64_t)in[2 * t + 1] << 32) | in[2 * t];
    uint32_t r = in[256 + t];
    out[t] = __builtin_amdgcn_qsad_pk_u16_u8(w, r, 0x0004000300020001ull);
    out[64 + t] = __builtin_amdgcn_mqsad_pk_u16_u8(w, r, 0x0004000300020001ull);
    out32[t] = __builtin_amdgcn_sad_u8(in[2 * t], r, 7u);
    out32[64 + t] = __builtin_amdgcn_udot4(in[2 * t], r, 7u, false);
    out32[128 + t] = __builtin_amdgcn_alignbyte(in[2 * t + 1], in[2 * t], (uint32_t)t);
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    std::vector<uint32_t> h(512);
    srand(7);
    for (auto& v : h) v = (uint32_t)rand() * 2654435761u ^ (uint32_t)rand();
    uint32_t* din; uint64_t* dout; long long* dcyc; uint32_t* dout32;
    CK(hipMalloc(&din, 512 * 4)); CK(hipMalloc(&dout, 8 * 1024 * 1024)); CK(hipMalloc(&dcyc, 8 * 65536)); CK(hipMalloc(&dout32, 4096));
    CK(hipMemcpy(din, h.data(), 512 * 4, hipMemcpyHostToDevice));

    // ---- semantics
    sem_kernel<<<1, 64>>>(din, dout, dout32);
    std::vector<uint64_t> o(128); std::vector<uint32_t> o32(192);
    CK(hipMemcpy(o.data(), dout, 128 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o32.data(), dout32, 192 * 4, hipMemcpyDeviceToHost));
    int bad_q = 0, bad_s = 0, bad_d = 0, bad_a = 0;
    for (int t = 0; t < 64; ++t) {
        uint8_t wb[8]; for (int k = 0; k < 4; ++k) { wb[k] = h[2 * t] >> (8 * k); wb[4 + k] = h[2 * t + 1] >> (8 * k); }
        uint8_t rb[4]; for (int k = 0; k < 4; ++k) rb[k] = h[256 + t] >> (8 * k);
        uint64_t want = 0;
        for (int i = 0; i < 4; ++i) {
            uint32_t s = i + 1;
            for (int k = 0; k < 4; ++k) s += abs((int)wb[i + k] - (int)rb[k]);
            want |= (uint64_t)(s & 0xffff) << (16 * i);
        }
        if (want != o[t]) { if (bad_q < 3) printf("qsad t=%d got %016llx want %016llx (w=%08x%08x r=%08x)\n", t, (unsigned long long)o[t], (unsigned long long)want, h[2*t+1], h[2*t], h[256+t]); ++bad_q; }
        uint32_t s = 7, d = 7;
        for (int k = 0; k < 4; ++k) { s += abs((int)wb[k] - (int)rb[k]); d += (uint32_t)wb[k] * rb[k]; }
        bad_s += s != o32[t]; bad_d += d != o32[64 + t];
        uint64_t both = ((uint64_t)h[2 * t + 1] << 32) | h[2 * t];
        uint32_t al = (uint32_t)(both >> (8 * (t & 3)));
        bad_a += al != o32[128 + t];
    }
    printf("semantics: qsad mismatches=%d sad=%d dot4=%d alignbyte=%d (of 64)\n", bad_q, bad_s, bad_d, bad_a);

    // ---- rates
    for (int w : {1, 2, 4, 8}) {
        run<5, false>("v_add_u32", din, dout, dcyc, w);
        run<0, false>("v_sad_u8", din, dout, dcyc, w);
        run<0, true>("v_sad_u8", din, dout, dcyc, w);
        run<1, false>("v_qsad_pk_u16_u8", din, dout, dcyc, w);
        run<1, true>("v_qsad_pk_u16_u8", din, dout, dcyc, w);
        run<2, false>("v_mqsad_pk_u16_u8", din, dout, dcyc, w);
        run<3, false>("v_dot4_u32_u8", din, dout, dcyc, w);
        run<3, true>("v_dot4_u32_u8", din, dout, dcyc, w);
        run<4, false>("v_alignbyte_b32", din, dout, dcyc, w);
        run<6, false>("v_sad_u16", din, dout, dcyc, w);
        run<7, false>("v_msad_u8", din, dout, dcyc, w);
    }
    return 0;
}
