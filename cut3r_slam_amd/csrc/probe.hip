// Measurement aid of bench.py (no reference counterpart): what the matrix pipe of THIS device sustains on fp16 operands that toggle.
// A bare MFMA loop -- operands in registers, 16 independent accumulator chains per wave, two waves per SIMD on every CU, no LDS, no
// memory traffic inside the loop -- stamped with s_memtime (shader clock) and s_memrealtime (100 MHz constant): the in-kernel clock the
// chip holds under MFMA load and the FLOP rate that goes with it.  The 2.5 PFLOP/s dense fp16 peak of MI355X_MICROARCH.md assumes
// 2.4 GHz; on N(0,1) operands the chip holds less (its "DVFS give-back" section), and bench.py reports both.
#include "common.h"
#include "../../include/cut3r_hip.h"

namespace {

__global__ __launch_bounds__(512) void mfma_probe_kernel(const h16* __restrict__ data, int nhalf, int iters, float* __restrict__ sink,
                                                         unsigned long long* __restrict__ stamps) {
    const int tid = threadIdx.x;
    half8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // (nhalf is a power of two >= 32768: every lane reads its own 128 bytes, wrapped)
        const unsigned base = ((blockIdx.x * 512u + tid) * 64u) & (unsigned)(nhalf - 1);
        a[i] = *reinterpret_cast<const half8_t*>(data + base + 8 * i);
        b[i] = *reinterpret_cast<const half8_t*>(data + base + 32 + 8 * i);
    }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[(size_t)blockIdx.x * 512 + tid] = s;
    if (tid == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

}  // namespace

extern "C" int cut3r_mfma_probe(const void* data, int nhalf, int iters, int grid, float* sink, unsigned long long* stamps, void* stream) {
    if (!data || !sink || !stamps || iters <= 0 || grid <= 0 || nhalf < 32768 || (nhalf & (nhalf - 1)) != 0) return CUT3R_ERR_ARG;
    if ((uintptr_t)data & 15) return CUT3R_ERR_ARG;
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, (const h16*)data, nhalf, iters, sink, stamps);
    return cut3r_check_launch();
}
