// TEST INFRASTRUCTURE - not part of the product, never shipped, never timed.
//
// A functional emulator of the gfx950 execution model, just wide enough for the kernels this repo generates: the
// generated HIP source of an integrator (ta.hip_source) is compiled for the HOST with this header in front of it and every
// work-item of a workgroup runs as a fibre (ucontext) of one OS thread. Cross-lane operations (DPP moves, ballots,
// shuffles, readfirstlane), the wave barrier of HY_WSYNC and __syncthreads() are rendezvous points of the fibres of a
// wavefront / workgroup, so LDS exchanges between the lanes behave as on the hardware as long as the control flow around
// them is wave-uniform - which is what the generators guarantee for the step loop (toolchain notes).
// tests/test_emulated_kernels.py uses it to compare the arithmetic of a generated kernel with the oracle in the
// authoring container, where there is no GPU; the GPU parity tests remain the proof for the hardware.
//
// What is NOT modelled: timing, bank conflicts, the exec mask inside divergent regions (cross-lane operations inside a
// divergent region abort), reduced-precision hardware functions (v_rcp_f64 is emulated as a correctly rounded quotient;
// the kernels refine it with Newton steps, so results agree to rounding).
#pragma once

#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace emu
{

struct uint3_t {
    unsigned x = 0, y = 0, z = 0;
};

struct fibre {
    ucontext_t ctx;
    std::vector<char> stack;
    uint3_t tid;
    bool done = false;
};

struct block_state {
    std::vector<fibre> fibres;
    ucontext_t main_ctx;
    fibre *cur = nullptr;
    uint3_t block_idx, grid_dim, block_dim;
    // Rendezvous bookkeeping: one (count, generation) pair per wavefront and one for the workgroup.
    std::vector<unsigned> wave_cnt, wave_gen;
    unsigned blk_cnt = 0, blk_gen = 0;
    // Exchange buffers.
    std::vector<std::uint64_t> xbuf; // [thread]
    int or_acc = 0, or_res = 0;
};

inline block_state *g_bs = nullptr;

inline void yield()
{
    swapcontext(&g_bs->cur->ctx, &g_bs->main_ctx);
}

inline unsigned lane_id()
{
    return g_bs->cur->tid.x & 63u;
}
inline unsigned wave_id()
{
    return g_bs->cur->tid.x >> 6;
}

inline void wave_sync()
{
    auto &b = *g_bs;
    const auto w = wave_id();
    const auto wave_size = std::min(64u, b.block_dim.x - w * 64u);
    const auto gen = b.wave_gen[w];
    if (++b.wave_cnt[w] == wave_size) {
        b.wave_cnt[w] = 0;
        ++b.wave_gen[w];
        return;
    }
    while (b.wave_gen[w] == gen) {
        yield();
    }
}

inline void block_sync()
{
    auto &b = *g_bs;
    const auto gen = b.blk_gen;
    if (++b.blk_cnt == b.block_dim.x) {
        b.blk_cnt = 0;
        ++b.blk_gen;
        return;
    }
    while (b.blk_gen == gen) {
        yield();
    }
}

inline int block_sync_or(int v)
{
    auto &b = *g_bs;
    b.or_acc |= (v != 0) ? 1 : 0;
    const auto gen = b.blk_gen;
    if (++b.blk_cnt == b.block_dim.x) {
        b.blk_cnt = 0;
        b.or_res = b.or_acc;
        b.or_acc = 0;
        ++b.blk_gen;
    } else {
        while (b.blk_gen == gen) {
            yield();
        }
    }
    const int r = b.or_res;
    // (Second rendezvous: nobody starts the next reduction before everybody has read this one.)
    block_sync();
    return r;
}

// Exchange of one 64-bit value between the lanes of a wavefront: every lane deposits, then reads the lane it wants.
template <typename F>
inline std::uint64_t wave_exchange(std::uint64_t v, F src_of)
{
    auto &b = *g_bs;
    const auto t = b.cur->tid.x;
    b.xbuf[t] = v;
    wave_sync();
    const unsigned src = src_of(t & 63u);
    const auto r = b.xbuf[(t & ~63u) + (src & 63u)];
    wave_sync();
    return r;
}

// DPP controls used by the generators: quad_perm (0x00-0xFF), row_shl/shr/ror (0x101-0x12F), row_mirror (0x140),
// row_half_mirror (0x141).
inline unsigned dpp_src(unsigned lane, unsigned ctrl)
{
    const unsigned row = lane & ~15u, in_row = lane & 15u;
    if (ctrl <= 0xFFu) {
        const unsigned q = lane & 3u;
        return (lane & ~3u) | ((ctrl >> (2u * q)) & 3u);
    }
    if (ctrl >= 0x101u && ctrl <= 0x10Fu) { // row_shl: lane i reads lane i + n
        return row | ((in_row + (ctrl & 15u)) & 15u);
    }
    if (ctrl >= 0x111u && ctrl <= 0x11Fu) { // row_shr: lane i reads lane i - n
        return row | ((in_row - (ctrl & 15u)) & 15u);
    }
    if (ctrl >= 0x121u && ctrl <= 0x12Fu) { // row_ror: rotate right by n: lane i reads lane i - n (mod 16)
        return row | ((in_row - (ctrl & 15u)) & 15u);
    }
    if (ctrl == 0x140u) {
        return row | (15u - in_row);
    }
    if (ctrl == 0x141u) {
        return (lane & ~7u) | (7u - (lane & 7u));
    }
    std::fprintf(stderr, "wave_emu: unsupported DPP control 0x%x\n", ctrl);
    std::abort();
}

inline int mov_dpp(int v, unsigned ctrl)
{
    return static_cast<int>(static_cast<std::uint32_t>(
        wave_exchange(static_cast<std::uint32_t>(v), [ctrl](unsigned lane) { return dpp_src(lane, ctrl); })));
}

inline std::uint64_t ballot(bool p)
{
    auto &b = *g_bs;
    const auto t = b.cur->tid.x;
    b.xbuf[t] = p ? 1u : 0u;
    wave_sync();
    std::uint64_t m = 0;
    const unsigned base = t & ~63u;
    for (unsigned i = 0; i < 64u && base + i < b.block_dim.x; ++i) {
        m |= (b.xbuf[base + i] & 1u) << i;
    }
    wave_sync();
    return m;
}

inline unsigned readfirstlane(unsigned v)
{
    return static_cast<unsigned>(wave_exchange(v, [](unsigned) { return 0u; }));
}

// Wave shuffles of any trivially copyable value of up to 8 bytes (int, unsigned long long, double).
template <typename T, typename F>
inline T shfl_any(T v, F src_of)
{
    static_assert(sizeof(T) <= 8u, "wave_emu: shuffle of a value wider than 64 bits");
    std::uint64_t bits = 0;
    std::memcpy(&bits, &v, sizeof(T));
    bits = wave_exchange(bits, src_of);
    T r;
    std::memcpy(&r, &bits, sizeof(T));
    return r;
}
template <typename T>
inline T shfl(T v, int src)
{
    return shfl_any(v, [src](unsigned) { return static_cast<unsigned>(src) & 63u; });
}
template <typename T>
inline T shfl_xor(T v, int m)
{
    return shfl_any(v, [m](unsigned lane) { return (lane ^ static_cast<unsigned>(m)) & 63u; });
}

inline int bpermute(int byte_addr, int v)
{
    return static_cast<int>(static_cast<std::uint32_t>(wave_exchange(
        static_cast<std::uint32_t>(v), [byte_addr](unsigned) { return (static_cast<unsigned>(byte_addr) >> 2) & 63u; })));
}

using kernel_fn = void (*)(const void *);
struct launch_ctx {
    kernel_fn fn;
    const void *args;
};
inline launch_ctx g_launch;

inline void fibre_entry()
{
    g_launch.fn(g_launch.args);
    g_bs->cur->done = true;
    swapcontext(&g_bs->cur->ctx, &g_bs->main_ctx);
}

// Runs a grid of workgroups one after the other (persistent kernels pull all their work through the device-side queue
// in the first workgroup - functionally the same thing).
inline void launch(kernel_fn fn, const void *args, unsigned grid, unsigned block)
{
    g_launch = {fn, args};
    for (unsigned bi = 0; bi < grid; ++bi) {
        block_state bs;
        bs.block_idx.x = bi;
        bs.grid_dim.x = grid;
        bs.block_dim.x = block;
        const unsigned n_waves = (block + 63u) / 64u;
        bs.wave_cnt.assign(n_waves, 0);
        bs.wave_gen.assign(n_waves, 0);
        bs.xbuf.assign(block, 0);
        bs.fibres.resize(block);
        g_bs = &bs;
        for (unsigned t = 0; t < block; ++t) {
            auto &f = bs.fibres[t];
            f.stack.resize(1u << 20);
            f.tid.x = t;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack.data();
            f.ctx.uc_stack.ss_size = f.stack.size();
            f.ctx.uc_link = &bs.main_ctx;
            makecontext(&f.ctx, fibre_entry, 0);
        }
        for (;;) {
            bool any = false;
            for (auto &f : bs.fibres) {
                if (f.done) {
                    continue;
                }
                any = true;
                bs.cur = &f;
                swapcontext(&bs.main_ctx, &f.ctx);
            }
            if (!any) {
                break;
            }
        }
        g_bs = nullptr;
    }
}

template <typename T>
inline T atomic_add(T *p, T v)
{
    const T old = *p;
    *p = old + v;
    return old;
}

} // namespace emu

// ---- the HIP / amdgcn vocabulary of the generated sources ----
#define __global__
#define __device__
#define __forceinline__ inline
#define __shared__ static
#define __constant__ static const
#define __launch_bounds__(...)
#define threadIdx (emu::g_bs->cur->tid)
#define blockIdx (emu::g_bs->block_idx)
#define gridDim (emu::g_bs->grid_dim)
#define blockDim (emu::g_bs->block_dim)

#define __builtin_amdgcn_wave_barrier() emu::wave_sync()
#define __builtin_amdgcn_s_barrier() emu::block_sync()
#define __syncthreads() emu::block_sync()
#define __syncthreads_or(x) emu::block_sync_or(x)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_mov_dpp(v, ctrl, rm, bm, bc) emu::mov_dpp((v), (ctrl))
#define __builtin_amdgcn_update_dpp(old, v, ctrl, rm, bm, bc) emu::mov_dpp((v), (ctrl))
#define __builtin_amdgcn_ballot_w64(p) emu::ballot(p)
#define __builtin_amdgcn_readfirstlane(v) emu::readfirstlane(v)
#define __builtin_amdgcn_ds_bpermute(a, v) emu::bpermute((a), (v))
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#define __builtin_amdgcn_rsq(x) (1.0 / std::sqrt(x))
#define __builtin_amdgcn_frexp_mant(x) emu_frexp_mant(x)
#define __builtin_amdgcn_frexp_exp(x) emu_frexp_exp(x)
#define __shfl(v, src, w) emu::shfl((v), (src))
#define __shfl_xor(v, m, w) emu::shfl_xor((v), (m))

inline double emu_frexp_mant(double x)
{
    int e;
    return std::frexp(x, &e);
}
inline int emu_frexp_exp(double x)
{
    int e;
    (void)std::frexp(x, &e);
    return e;
}
inline int __double2loint(double x)
{
    std::uint64_t b;
    std::memcpy(&b, &x, 8);
    return static_cast<int>(static_cast<std::uint32_t>(b));
}
inline int __double2hiint(double x)
{
    std::uint64_t b;
    std::memcpy(&b, &x, 8);
    return static_cast<int>(static_cast<std::uint32_t>(b >> 32));
}
inline double __hiloint2double(int hi, int lo)
{
    const std::uint64_t b = (static_cast<std::uint64_t>(static_cast<std::uint32_t>(hi)) << 32) | static_cast<std::uint32_t>(lo);
    double r;
    std::memcpy(&r, &b, 8);
    return r;
}
inline double __longlong_as_double(long long v)
{
    double r;
    std::memcpy(&r, &v, 8);
    return r;
}
inline long long __double_as_longlong(double v)
{
    long long r;
    std::memcpy(&r, &v, 8);
    return r;
}
inline unsigned atomicAdd(unsigned *p, unsigned v)
{
    return emu::atomic_add(p, v);
}
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v)
{
    return emu::atomic_add(p, v);
}
inline unsigned atomicOr(unsigned *p, unsigned v)
{
    const unsigned old = *p;
    *p = old | v;
    return old;
}
inline unsigned atomicMax(unsigned *p, unsigned v)
{
    const unsigned old = *p;
    *p = old > v ? old : v;
    return old;
}
using std::exp;
using std::fabs;
using std::fmax;
using std::fmin;
using std::log;
using std::sqrt;
using std::ldexp;
