// Issue-rate microbenchmarks for gfx950: how many cycles does one wave need per instruction of each kind,
// alone on its SIMD and with a partner wave?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define FMA8 \
  "v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n" \
  "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n"
#define FMA8_X(X) \
  "v_fma_f64 %0, %8, %9, %0\n" X "v_fma_f64 %1, %8, %9, %1\n" X "v_fma_f64 %2, %8, %9, %2\n" X "v_fma_f64 %3, %8, %9, %3\n" X \
  "v_fma_f64 %4, %8, %9, %4\n" X "v_fma_f64 %5, %8, %9, %5\n" X "v_fma_f64 %6, %8, %9, %6\n" X "v_fma_f64 %7, %8, %9, %7\n" X
#define FMA_DEP8 \
  "v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %0, %8, %9, %0\n" \
  "v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %0, %8, %9, %0\n"
#define FMA_DEP2_8 \
  "v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n" \
  "v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n"
#define FMA_DEP3_8 \
  "v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %0, %8, %9, %0\n" \
  "v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n"

#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y), "v"(addr), "v"(tmp)

template <int K>
__global__ void __launch_bounds__(512) kern(double *out, long long *cyc, int iters, int ldspad)
{
    extern __shared__ double lds[];
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    double x = 1.0000001, y = 1e-9;
    unsigned addr = (threadIdx.x * 8u) & 4095u;
    unsigned tmp = 0;
    lds[threadIdx.x] = 1.0;
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if constexpr (K == 0) asm volatile(REP8(FMA8) : OPS);
        if constexpr (K == 1) asm volatile(REP8(FMA8_X("s_nop 0\n")) : OPS);
        if constexpr (K == 2) asm volatile(REP8(FMA8_X("v_mov_b32 %11, %10\n")) : OPS);
        if constexpr (K == 3) asm volatile(REP8(FMA8_X("v_accvgpr_write_b32 a0, %10\n")) : OPS : "a0");
        if constexpr (K == 4) asm volatile(REP8(FMA_DEP8) : OPS);
        if constexpr (K == 5) asm volatile(REP8(FMA_DEP2_8) : OPS);
        if constexpr (K == 6) asm volatile(REP8(FMA_DEP3_8) : OPS);
        if constexpr (K == 7) asm volatile(REP8(FMA8_X("s_mov_b32 s20, 0x40220000\n")) : OPS : "s20");
        if constexpr (K == 8) asm volatile(REP8(FMA8_X("s_nop 0\n s_nop 0\n")) : OPS);
        if constexpr (K == 9) asm volatile(REP8(FMA8_X("v_mov_b32 %11, %10\n v_mov_b32 %11, %10\n")) : OPS);
        if constexpr (K == 10) { // fma + independent ds_read_b64 (result unused until the end of the block)
            double r0;
            asm volatile(REP8(FMA8_X("ds_read_b64 %12, %10\n")) "s_waitcnt lgkmcnt(0)\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y), "v"(addr), "v"(tmp), "v"(r0));
        }
        if constexpr (K == 11) asm volatile(REP8(FMA8_X("v_add_f64 %0, %8, %9\n")) : OPS);  // pure fp64 mix
        if constexpr (K == 12) asm volatile(REP8(FMA8_X("ds_write_b64 %10, %8\n")) "s_waitcnt lgkmcnt(0)\n" : OPS);
        if constexpr (K == 13) asm volatile(REP8(FMA8_X("s_waitcnt lgkmcnt(0)\n")) : OPS);
        if constexpr (K == 14) asm volatile(REP8(FMA8_X("v_cndmask_b32 %11, %10, %10, vcc\n")) : OPS);
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + tmp;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int K>
void run(const char *name, int per_iter_fma, int per_iter_other)
{
    for (int bs : {256, 512}) {
        int nb = 256, iters = 2000;
        double *out; long long *cyc;
        hipMalloc(&out, sizeof(double) * nb * bs);
        hipMalloc(&cyc, sizeof(long long) * nb * 8);
        size_t lds = 100 * 1024; // one block per CU
        hipFuncSetAttribute((const void *)kern<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        kern<K><<<nb, bs, lds>>>(out, cyc, 10, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        kern<K><<<nb, bs, lds>>>(out, cyc, iters, 0);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(nb * bs / 64);
        hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= h.size();
        double n_fma = (double)iters * per_iter_fma;
        // clock64 = s_memtime at 100 MHz? report both wall-derived cycles at 2.4 GHz and counter ticks
        printf("%-28s waves/SIMD=%d  ticks/fma=%.2f  ns/fma(wall)=%.3f  -> cycles@2.4GHz per fma slot=%.2f (others per fma: %.2f)\n", name, bs / 256, avg / n_fma,
               ms * 1e6 / n_fma, ms * 1e6 / n_fma * 2.4, (double)per_iter_other / per_iter_fma);
        hipFree(out); hipFree(cyc);
    }
}

int main()
{
    run<0>("fma x8 indep", 64, 0);
    run<4>("fma dep chain 1", 64, 0);
    run<5>("fma dep chains 2", 64, 0);
    run<6>("fma dep chains 3", 64, 0);
    run<1>("fma + s_nop", 64, 64);
    run<8>("fma + 2 s_nop", 64, 128);
    run<7>("fma + s_mov_b32", 64, 64);
    run<13>("fma + s_waitcnt", 64, 64);
    run<2>("fma + v_mov_b32", 64, 64);
    run<9>("fma + 2 v_mov_b32", 64, 128);
    run<14>("fma + v_cndmask", 64, 64);
    run<3>("fma + v_accvgpr_write", 64, 64);
    run<11>("fma + v_add_f64", 64, 64);
    run<10>("fma + ds_read_b64", 64, 64);
    run<12>("fma + ds_write_b64", 64, 64);
    return 0;
}
