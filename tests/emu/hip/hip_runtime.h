// TEST INFRASTRUCTURE ONLY -- a tiny host-side emulator of the HIP execution model.
//
// The product kernels in action-detection_amd/csrc/*.hip are written for gfx950 only.  To
// exercise their index logic (gather loaders, LDS tiling, MFMA fragment maps, epilogues) in
// the CPU-only test tier, the test build compiles the SAME sources with the host clang and
// this header shadowing <hip/hip_runtime.h> (-I tests/emu).  One workgroup runs at a time;
// each work-item is a ucontext fiber; __syncthreads / wave shuffles / MFMA are rendezvous
// points resolved by a cooperative scheduler (deadlock -> abort with a message).
//
// The MFMA emulation implements the documented gfx950 lane->element maps
// (/opt/skills/guides/cdna_hip_programming.md section 3): for v_mfma_f32_32x32x2_f32 lane l
// supplies A[i=l&31][k=l>>5], B[k=l>>5][j=l&31] and owns D[(r&3)+8*(r>>2)+4*(l>>5)][l&31].
#pragma once
#include <ucontext.h>

#include <cassert>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __HIP_DEVICE_COMPILE__ 1   /* the emulator executes device code */
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 {
    unsigned int x, y;
};
struct float4 {
    float x, y, z, w;
};
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct float2 {
    float x, y;
};

typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }

namespace emu {

enum Wait { W_NONE = 0, W_WAVE = 1, W_BLOCK = 2 };

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    dim3 tid;
    int lin = 0, lane = 0, wave = 0;
    bool done = false;
    int wait = W_NONE;
    uint64_t wave_seq = 0;   // number of wave ops this lane has entered
    uint64_t block_seq = 0;  // number of block barriers this thread has entered
};

struct WaveX {
    // double-buffered exchange area: 4 dwords per lane
    uint32_t buf[2][64][8];
};

struct BlockState {
    std::vector<Fiber> fibers;
    std::vector<WaveX> waves;
    dim3 bid, bdim, gdim;
    ucontext_t sched;
    std::function<void()> body;
};

extern BlockState* g_blk;
extern Fiber* g_cur;

inline void yield_to_sched() { swapcontext(&g_cur->ctx, &g_blk->sched); }

inline bool can_run(const Fiber& f) {
    BlockState& b = *g_blk;
    if (f.wait == W_NONE) return true;
    if (f.wait == W_WAVE) {
        for (auto& o : b.fibers)
            if (o.wave == f.wave && !o.done && o.wave_seq < f.wave_seq) return false;
        return true;
    }
    for (auto& o : b.fibers)
        if (!o.done && o.block_seq < f.block_seq) return false;
    return true;
}

inline void fiber_entry() {
    g_blk->body();
    g_cur->done = true;
    swapcontext(&g_cur->ctx, &g_blk->sched);
}

inline char* stack_pool(unsigned i, size_t bytes) {
    static std::vector<char*> pool;
    if (pool.size() <= i) pool.resize(i + 1, nullptr);
    if (!pool[i]) pool[i] = (char*)malloc(bytes);
    return pool[i];
}

inline void run_block(BlockState& b) {
    g_blk = &b;
    const unsigned n = b.bdim.x * b.bdim.y * b.bdim.z;
    b.fibers.resize(n);
    b.waves.assign((n + 63) / 64, WaveX());
    const size_t stk = 512 * 1024;
    for (unsigned i = 0; i < n; ++i) {
        Fiber& f = b.fibers[i];
        f.lin = i;
        f.tid = dim3(i % b.bdim.x, (i / b.bdim.x) % b.bdim.y, i / (b.bdim.x * b.bdim.y));
        f.lane = i & 63;
        f.wave = i >> 6;
        f.done = false;
        f.wait = W_NONE;
        f.wave_seq = f.block_seq = 0;
        f.stack = stack_pool(i, stk);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = stk;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
    for (;;) {
        bool all_done = true, progressed = false;
        for (unsigned i = 0; i < n; ++i) {
            Fiber& f = b.fibers[i];
            if (f.done) continue;
            all_done = false;
            if (!can_run(f)) continue;
            f.wait = W_NONE;
            g_cur = &f;
            swapcontext(&b.sched, &f.ctx);
            progressed = true;
        }
        if (all_done) break;
        if (!progressed) {
            fprintf(stderr, "[hip-emu] deadlock: divergent barrier / wave op in block (%u,%u,%u)\n", b.bid.x,
                    b.bid.y, b.bid.z);
            abort();
        }
    }
}

// ---- rendezvous primitives ----
inline void block_barrier() {
    g_cur->block_seq++;
    g_cur->wait = W_BLOCK;
    yield_to_sched();
}

// deposit up to 8 dwords, wait for the whole wave, return pointer to the wave's slot array
inline uint32_t (*wave_exchange(const uint32_t* v, int n))[8] {
    Fiber* f = g_cur;
    WaveX& w = g_blk->waves[f->wave];
    const int slot = (int)(f->wave_seq & 1);
    for (int i = 0; i < n; ++i) w.buf[slot][f->lane][i] = v[i];
    f->wave_seq++;
    f->wait = W_WAVE;
    yield_to_sched();
    return w.buf[slot];
}

inline int wave_lanes() {
    // number of live lanes in the current wave (last wave of a block may be partial)
    const unsigned n = g_blk->bdim.x * g_blk->bdim.y * g_blk->bdim.z;
    const unsigned base = (unsigned)g_cur->wave * 64u;
    return (int)((n - base) < 64u ? (n - base) : 64u);
}

template <class T>
inline T shfl_idx(T v, int src) {
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    uint32_t u;
    memcpy(&u, &v, 4);
    auto* buf = wave_exchange(&u, 1);
    uint32_t r = buf[src & 63][0];
    if ((src & 63) >= wave_lanes()) r = u;
    T o;
    memcpy(&o, &r, 4);
    return o;
}

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

inline f32x16_t mfma_32x32x2(float a, float b, f32x16_t c) {
    uint32_t u[2];
    memcpy(&u[0], &a, 4);
    memcpy(&u[1], &b, 4);
    auto* buf = wave_exchange(u, 2);
    const int lane = g_cur->lane;
    const int col = lane & 31, hi = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            memcpy(&av, &buf[k * 32 + row][0], 4);
            memcpy(&bv, &buf[k * 32 + col][1], 4);
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    return c;
}

inline f32x4_t mfma_16x16x4(float a, float b, f32x4_t c) {
    uint32_t u[2];
    memcpy(&u[0], &a, 4);
    memcpy(&u[1], &b, 4);
    auto* buf = wave_exchange(u, 2);
    const int lane = g_cur->lane;
    const int col = lane & 15, grp = lane >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = grp * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            memcpy(&av, &buf[k * 16 + row][0], 4);
            memcpy(&bv, &buf[k * 16 + col][1], 4);
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    return c;
}

// v_mfma_f32_32x32x16_bf16: lane l supplies A[i = l & 31][k = 8 (l >> 5) + 0..7] (8 bf16 in 4 dwords, k even in
// the low half) and likewise B[k][j = l & 31]; the C/D layout is that of the 32x32x2 f32 instruction.
template <class V>
inline f32x16_t mfma_32x32x16_bf16(V a, V b, f32x16_t c) {
    static_assert(sizeof(V) == 16, "8 x bf16 operands");
    uint32_t u[8];
    memcpy(&u[0], &a, 16);
    memcpy(&u[4], &b, 16);
    auto* buf = wave_exchange(u, 8);
    const int lane = g_cur->lane;
    const int col = lane & 31, hi = lane >> 5;
    auto elem = [&](int src_lane, int base, int kk) {
        const uint32_t d = buf[src_lane][base + (kk >> 1)];
        const uint32_t bits = (kk & 1) ? (d & 0xFFFF0000u) : (d << 16);
        float f;
        memcpy(&f, &bits, 4);
        return f;
    };
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 16; ++k)
            acc = fmaf(elem((k >> 3) * 32 + row, 0, k & 7), elem((k >> 3) * 32 + col, 4, k & 7), acc);
        c[r] = acc;
    }
    return c;
}

// v_mfma_f32_32x32x16_f16: same operand / result maps as the bf16 form, elements are IEEE halves
template <class V>
inline f32x16_t mfma_32x32x16_f16(V a, V b, f32x16_t c) {
    static_assert(sizeof(V) == 16, "8 x f16 operands");
    uint32_t u[8];
    memcpy(&u[0], &a, 16);
    memcpy(&u[4], &b, 16);
    auto* buf = wave_exchange(u, 8);
    const int lane = g_cur->lane;
    const int col = lane & 31, hi = lane >> 5;
    auto elem = [&](int src_lane, int base, int kk) {
        const uint32_t d = buf[src_lane][base + (kk >> 1)];
        const uint16_t h = (uint16_t)((kk & 1) ? (d >> 16) : (d & 0xFFFFu));
        _Float16 f;
        memcpy(&f, &h, 2);
        return (float)f;
    };
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 16; ++k)
            acc = fmaf(elem((k >> 3) * 32 + row, 0, k & 7), elem((k >> 3) * 32 + col, 4, k & 7), acc);
        c[r] = acc;
    }
    return c;
}

template <class K, class... Args>
inline void launch(K kernel, dim3 grid, dim3 block, Args... args) {
    BlockState b;
    b.gdim = grid;
    b.bdim = block;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                b.bid = dim3(x, y, z);
                b.body = [&]() { kernel(args...); };
                run_block(b);
            }
}

}  // namespace emu

#define threadIdx (emu::g_cur->tid)
#define blockIdx (emu::g_blk->bid)
#define blockDim (emu::g_blk->bdim)
#define gridDim (emu::g_blk->gdim)

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch(kernel, dim3(grid), dim3(block), __VA_ARGS__)

static inline void __syncthreads() { emu::block_barrier(); }
template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width;
    return emu::shfl_idx(v, emu::g_cur->lane ^ mask);
}
template <class T>
static inline T __shfl(T v, int src, int width = 64) {
    (void)width;
    return emu::shfl_idx(v, src);
}
template <class T>
static inline T __shfl_down(T v, int delta, int width = 64) {
    (void)width;
    int src = emu::g_cur->lane + delta;
    return emu::shfl_idx(v, src > 63 ? emu::g_cur->lane : src);
}
// v_permlane32_swap_b32 vdst, src (gfx950): lanes 32-63 of vdst swap with lanes 0-31 of src; returns {vdst', src'}
typedef uint32_t emu_u32x2_swap __attribute__((ext_vector_type(2)));
static inline emu_u32x2_swap __builtin_amdgcn_permlane32_swap(uint32_t vdst, uint32_t src, bool, bool) {
    uint32_t u[2] = {vdst, src};
    auto* buf = emu::wave_exchange(u, 2);
    const int lane = emu::g_cur->lane;
    emu_u32x2_swap r;
    if (lane < 32) {
        r[0] = vdst;
        r[1] = buf[lane + 32][0];
    } else {
        r[0] = buf[lane - 32][1];
        r[1] = src;
    }
    return r;
}
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline float atomicAdd(float* p, float v) {
    float o = *p;
    *p = o + v;
    return o;
}
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
    unsigned long long o = *p;
    *p = o + v;
    return o;
}
static inline int atomicAdd(int* p, int v) {
    int o = *p;
    *p = o + v;
    return o;
}
static inline unsigned int atomicMax(unsigned int* p, unsigned int v) {
    unsigned int o = *p;
    if (v > o) *p = v;
    return o;
}
static inline float __fdividef(float a, float b) { return a / b; }
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu::mfma_32x32x2(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu::mfma_16x16x4(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu::mfma_32x32x16_bf16(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu::mfma_32x32x16_f16(a, b, c)
// v_perm_b32: selector bytes 0-3 pick bytes of the SECOND operand, 4-7 bytes of the first
static inline uint32_t __builtin_amdgcn_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
    const uint64_t both = ((uint64_t)s0 << 32) | s1;
    uint32_t out = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t q = (sel >> (8 * i)) & 0xFF;
        const uint32_t byte = q < 8 ? (uint32_t)((both >> (8 * q)) & 0xFF) : (q == 12 ? 0u : 0xFFu);
        out |= byte << (8 * i);
    }
    return out;
}
#define __builtin_amdgcn_s_getreg(x) 0
#define __builtin_amdgcn_s_memrealtime() 0ull
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)   /* only used on wave-uniform values */
struct __amdgpu_buffer_rsrc_t {
    const char* base;
    uint32_t bytes;
};
// Allocation registry (tests/conftest.py registers every planes tensor it sees created): a buffer descriptor whose base lies inside
// a registered allocation must END inside it too.  On the GPU the range check of a descriptor is the only thing between a per-lane
// gather and the memory behind a tensor; a descriptor that claims more bytes than its allocation holds reads whatever follows it
// (round 4: 60 KB past the last image of a slice at the end of its tensor -- found on the GPU, invisible here).  Violations are
// counted, the first one described; the test fixture fails the test that caused them.
namespace emu {
void descriptor_check(const void* base, uint32_t bytes);
}
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int bytes, int) {
    emu::descriptor_check(p, (uint32_t)bytes);
    return __amdgpu_buffer_rsrc_t{(const char*)p, (uint32_t)bytes};
}
typedef unsigned int emu_u32x4 __attribute__((ext_vector_type(4)));
// raw buffer loads: offsets at or beyond num_records return zero (hardware range check)
static inline unsigned int __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff,
                                                               int) {
    const uint64_t o = (uint64_t)voff + soff;
    unsigned int v = 0;
    if (o + 4 <= r.bytes) memcpy(&v, r.base + o, 4);
    return v;
}
static inline unsigned char __builtin_amdgcn_raw_buffer_load_b8(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff,
                                                               int) {
    const uint64_t o = (uint64_t)voff + soff;
    return (o + 1 <= r.bytes) ? (unsigned char)r.base[o] : (unsigned char)0;
}
static inline emu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff,
                                                              int) {
    // multi-dword raw loads are range-checked dword by dword
    const uint64_t o = (uint64_t)voff + soff;
    unsigned int w[4] = {0, 0, 0, 0};
    for (int d = 0; d < 4; ++d)
        if (o + 4 * d + 4 <= r.bytes) memcpy(&w[d], r.base + o + 4 * d, 4);
    return emu_u32x4{w[0], w[1], w[2], w[3]};
}
typedef unsigned int emu_u32x2 __attribute__((ext_vector_type(2)));
static inline emu_u32x2 __builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, int) {
    const uint64_t o = (uint64_t)voff + soff;
    unsigned int w[2] = {0, 0};
    for (int d = 0; d < 2; ++d)
        if (o + 4 * d + 4 <= r.bytes) memcpy(&w[d], r.base + o + 4 * d, 4);
    return emu_u32x2{w[0], w[1]};
}
static inline unsigned short __builtin_amdgcn_raw_buffer_load_b16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff,
                                                                  int) {
    const uint64_t o = (uint64_t)voff + soff;
    unsigned short v = 0;
    if (o + 2 <= r.bytes) memcpy(&v, r.base + o, 2);
    return v;
}
// raw buffer store: offsets at or beyond num_records are dropped
static inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned int v, __amdgpu_buffer_rsrc_t r, uint32_t voff,
                                                         uint32_t soff, int) {
    const uint64_t o = (uint64_t)voff + soff;
    if (o + 4 <= r.bytes) memcpy(const_cast<char*>(r.base) + o, &v, 4);
}
static inline void __builtin_amdgcn_raw_buffer_store_b64(emu_u32x2 v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff,
                                                         int) {
    const uint64_t o = (uint64_t)voff + soff;
    for (int d = 0; d < 2; ++d)
        if (o + 4 * d + 4 <= r.bytes) {
            const unsigned int w = v[d];
            memcpy(const_cast<char*>(r.base) + o + 4 * d, &w, 4);
        }
}
static inline void __builtin_amdgcn_raw_buffer_store_b128(emu_u32x4 v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff,
                                                          int) {
    const uint64_t o = (uint64_t)voff + soff;
    for (int d = 0; d < 4; ++d)
        if (o + 4 * d + 4 <= r.bytes) {
            const unsigned int w = v[d];
            memcpy(const_cast<char*>(r.base) + o + 4 * d, &w, 4);
        }
}
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) {
    const float lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
}
static inline int atomicOr(int* p, int v) {
    int o = *p;
    *p = o | v;
    return o;
}
// ds_read_b64_tr_b16 (gfx950 LDS transpose read): every lane supplies the address of 4 consecutive 16-bit elements (one
// row piece); inside each group of 16 lanes, lanes 4j..4j+3 supply row j (16 elements), and lane c of the group receives
// column c of rows 0..3 (element j = row j).
static inline emu_u32x2 emu_ds_read_tr16_b64(const void* addr) {
    uint32_t u[2];
    memcpy(u, addr, 8);
    auto* buf = emu::wave_exchange(u, 2);
    const int lane = emu::g_cur->lane, grp = lane & ~15, c = lane & 15;
    unsigned short e[4];
    for (int j = 0; j < 4; ++j) {
        const uint32_t* src = buf[grp + 4 * j + (c >> 2)];
        const uint32_t d = src[(c & 3) >> 1];
        e[j] = (unsigned short)((c & 1) ? (d >> 16) : (d & 0xFFFFu));
    }
    return emu_u32x2{(uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16)};
}
#define SSN_DS_READ_TR16_B64(ptr) emu_ds_read_tr16_b64(ptr)
#define SSN_DS_READ_TR16_B64_AT(dst, ptr, imm) ((dst) = emu_ds_read_tr16_b64((const char*)(ptr) + (imm)))
// LDS-DMA: every lane deposits `size` bytes at (wave-uniform lds base) + lane * size
#define SSN_LDS_PTR(p) ((void*)(p))
#define SSN_CONST_PTR(T, p) ((const T*)(p))
#define SSN_WAIT_VMCNT(n) ((void)0)
#define SSN_STORE_DATA_GUARD(v) ((void)0)
#define PL_LO_PAIR(dst, v0, v1, hi) ((dst) = f16_pair_rne((v0) - f16_pair_lo(hi), (v1) - f16_pair_hi(hi)))
#define SSN_WAIT_LGKM0() ((void)0)
static inline void __builtin_amdgcn_raw_ptr_buffer_load_lds(__amdgpu_buffer_rsrc_t r, void* lds, int size, uint32_t voff,
                                                            uint32_t soff, int imm, int) {
    const uint64_t o = (uint64_t)voff + soff + (uint32_t)imm;
    char* dst = (char*)lds + (size_t)emu::g_cur->lane * size;
    for (int d = 0; d < size; d += 4) {       // range check dword by dword
        if (o + d + 4 <= r.bytes)
            memcpy(dst + d, r.base + o + d, 4);
        else
            memset(dst + d, 0, 4);
    }
}
static inline void __builtin_amdgcn_s_barrier() { emu::block_barrier(); }
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(m, n, id) ((void)0)
