// Issue-rate microbenchmark v2 (inline asm, long kernels, in-kernel clock):
//   hipcc --offload-arch=gfx950 -O3 valu_rates2.hip -o valu_rates2 && ./valu_rates2
// Reports, per instruction and waves/SIMD: shader cycles per wave-instruction per SIMD
// (s_memtime ticks) and the shader clock derived from s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 4096;
constexpr int NACC = 16;

#define OP3_32(name) asm volatile(name " %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y))
#define OP3_32S(name) asm volatile(name " %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(sy))

template <int OP>
__global__ void __launch_bounds__(1024) rate_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, long long* stamps, uint32_t sy)
{
    uint32_t x = in[threadIdx.x & 63], y = in[64 + (threadIdx.x & 63)];
    uint64_t w = ((uint64_t)in[128 + (threadIdx.x & 63)] << 32) | x;
    uint32_t a[NACC];
    uint64_t q[NACC];
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 m[4];
    for (int i = 0; i < NACC; ++i) { a[i] = i + threadIdx.x; q[i] = i; }
    for (int i = 0; i < 4; ++i) m[i] = u4{0, 0, 0, 0};
    long long t0 = __builtin_amdgcn_s_memtime();
    long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (OP == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
            if (OP == 1) OP3_32("v_sad_u8");
            if (OP == 2) OP3_32S("v_sad_u8");
            if (OP == 3) asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(q[i]) : "v"(w), "v"(y));
            if (OP == 4) asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(q[i]) : "v"(w), "s"(sy));
            if (OP == 5) asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0" : "+v"(m[i & 3]) : "v"(w), "v"(y));
            if (OP == 6) OP3_32("v_dot4_u32_u8");
            if (OP == 7) OP3_32("v_alignbyte_b32");
            if (OP == 8) OP3_32("v_perm_b32");
            if (OP == 9) OP3_32("v_fma_f32");
            if (OP == 10) OP3_32("v_add3_u32");
            if (OP == 11) OP3_32("v_mad_u32_u24");
            if (OP == 12) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
            if (OP == 13) OP3_32("v_lshl_or_b32");
            if (OP == 14) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
            if (OP == 15) OP3_32("v_sad_u16");
            if (OP == 16) OP3_32("v_bfe_u32");
            if (OP == 17) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(a[i]) : "v"(x));
            if (OP == 18) OP3_32("v_and_or_b32");
            if (OP == 19) asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(q[i]) : "v"(w), "v"(y));
            if (OP == 20) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sy), "v"(y));
            if (OP == 21) asm volatile("v_min3_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    long long r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t s = 0;
    for (int i = 0; i < NACC; ++i) s += a[i] + (uint32_t)q[i] + (uint32_t)(q[i] >> 32);
    for (int i = 0; i < 4; ++i) s += m[i].x + m[i].y + m[i].z + m[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        int wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        stamps[2 * wv] = t1 - t0; stamps[2 * wv + 1] = r1 - r0;
    }
}

template <int OP>
void run(const char* name, const uint32_t* din, uint32_t* dout, long long* dst, int wps)
{
    int threads = 64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps;
    int wg_per_cu = (64 * 4 * wps) / threads;
    int blocks = 256 * wg_per_cu;
    int nwaves = blocks * threads / 64;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<OP><<<blocks, threads>>>(din, dout, dst, 0x12345678u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    rate_kernel<OP><<<blocks, threads>>>(din, dout, dst, 0x12345678u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> st(2 * nwaves);
    CK(hipMemcpy(st.data(), dst, sizeof(long long) * 2 * nwaves, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0; for (int i = 0; i < nwaves; ++i) { cyc += st[2 * i]; rt += st[2 * i + 1]; }
    cyc /= nwaves; rt /= nwaves;
    double clock_ghz = cyc / (rt * 10.0);               // rt ticks are 10 ns
    double instr = (double)ITERS * NACC;
    printf("%-20s w/simd=%d  wall=%.3f ms  cyc/wave=%.0f  clock=%.2f GHz  cyc per instr per SIMD=%.2f  (wall-derived ns/instr/SIMD=%.3f)\n",
           name, wps, ms, cyc, clock_ghz, cyc / (instr * wps), ms * 1e6 / (instr * wps));
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d\n", prop.gcnArchName, prop.multiProcessorCount);
    std::vector<uint32_t> h(512); srand(7);
    for (auto& v : h) v = (uint32_t)rand() * 2654435761u ^ (uint32_t)rand();
    uint32_t* din; uint32_t* dout; long long* dst;
    CK(hipMalloc(&din, 2048)); CK(hipMalloc(&dout, 4 * 1024 * 1024)); CK(hipMalloc(&dst, 16 * 65536));
    CK(hipMemcpy(din, h.data(), 2048, hipMemcpyHostToDevice));
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_add_u32", din, dout, dst, w);
        run<14>("v_xor_b32", din, dout, dst, w);
        run<12>("v_min_u32", din, dout, dst, w);
        run<17>("v_pk_add_u16", din, dout, dst, w);
        run<9>("v_fma_f32", din, dout, dst, w);
        run<10>("v_add3_u32", din, dout, dst, w);
        run<21>("v_min3_u32", din, dout, dst, w);
        run<11>("v_mad_u32_u24", din, dout, dst, w);
        run<13>("v_lshl_or_b32", din, dout, dst, w);
        run<18>("v_and_or_b32", din, dout, dst, w);
        run<16>("v_bfe_u32", din, dout, dst, w);
        run<7>("v_alignbyte_b32", din, dout, dst, w);
        run<8>("v_perm_b32", din, dout, dst, w);
        run<1>("v_sad_u8", din, dout, dst, w);
        run<2>("v_sad_u8 (sgpr src1)", din, dout, dst, w);
        run<20>("v_sad_u8 (sgpr src0)", din, dout, dst, w);
        run<15>("v_sad_u16", din, dout, dst, w);
        run<6>("v_dot4_u32_u8", din, dout, dst, w);
        run<3>("v_qsad_pk_u16_u8", din, dout, dst, w);
        run<4>("v_qsad_pk (sgpr ref)", din, dout, dst, w);
        run<19>("v_mqsad_pk_u16_u8", din, dout, dst, w);
        run<5>("v_mqsad_u32_u8", din, dout, dst, w);
    }
    return 0;
}
