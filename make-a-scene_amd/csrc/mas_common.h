// Shared device/host helpers for libmas_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/mas_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---- error plumbing (host) -------------------------------------------------------
void mas_set_error(const char* fmt, ...);
#define MAS_FAIL(code, ...) do { mas_set_error(__VA_ARGS__); return (code); } while (0)
void mas_note_kernel(const char* name);      // which kernel the last successful launch of this process was (mas_last_kernel)
#define MAS_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) MAS_FAIL(MAS_ELAUNCH, "%s: launch failed: %s", name, hipGetErrorString(e_)); mas_note_kernel(name); } while (0)

// a stale error left in this thread by another library must not be blamed on our launch
#define MAS_ENTER() do { (void)hipGetLastError(); } while (0)

int mas_num_cus();   // compute units of the CURRENT device (cached per device)

// One process per GPU is the contract (include/mas_hip.h), but a second device in the same process (nn.DataParallel,
// reference train.py:177) must still launch correctly: per-function attributes such as the dynamic-LDS limit are set
// once PER DEVICE.  `mask` holds one bit per device ordinal; the attribute call is idempotent, so a race only repeats it.
#include <atomic>
typedef std::atomic<unsigned long long> mas_devmask_t;
static inline bool mas_attr_needed(mas_devmask_t& mask, unsigned long long* bit) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    *bit = 1ull << (dev & 63);
    return (mask.load(std::memory_order_acquire) & *bit) == 0;
}
static inline void mas_attr_done(mas_devmask_t& mask, unsigned long long bit) { mask.fetch_or(bit, std::memory_order_release); }
static inline int mas_roundup(int a, int b) { return (a + b - 1) / b * b; }
static inline int mas_cdiv(int a, int b) { return (a + b - 1) / b; }
// tuning / A-B knobs are read from the environment ONCE per process (not on every launch)
static inline int mas_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline size_t mas_esize(int dtype) { return dtype == MAS_BF16 ? 2 : 4; }

// ---- bf16 <-> f32 (round-to-nearest-even, same as torch's .to(bfloat16)) ---------
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)f; }

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int DT = MAS_F32;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int DT = MAS_BF16;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return (float)*p; }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = (bf16_t)v; }
};

// 8 consecutive elements as a register vector
template <typename T> struct Vec8;
template <> struct Vec8<float> { typedef f32x8 type; };
template <> struct Vec8<bf16_t> { typedef bf16x8 type; };

template <typename T>
__device__ __forceinline__ typename Vec8<T>::type ld8(const T* p) {   // p 16B (bf16) / 32B (f32) aligned
    return *reinterpret_cast<const typename Vec8<T>::type*>(p);
}
template <typename T>
__device__ __forceinline__ void st8(T* p, typename Vec8<T>::type v) {
    *reinterpret_cast<typename Vec8<T>::type*>(p) = v;
}
template <typename T>
__device__ __forceinline__ typename Vec8<T>::type zero8() {
    typename Vec8<T>::type v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (T)0.0f;
    return v;
}

// v_exp_f32 + v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division sequence: this runs once per staged element
__device__ __forceinline__ float silu_f(float u) { return u * __builtin_amdgcn_rcpf(1.0f + __expf(-u)); }
// two at a time on the packed-fp32 instructions (v_pk_mul / v_pk_add around the two transcendentals each): bitwise the same values
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ f32x2 silu2_f(f32x2 u) {
    f32x2 t = u * -1.4426950408889634f;
    t[0] = __builtin_amdgcn_exp2f(t[0]); t[1] = __builtin_amdgcn_exp2f(t[1]);
    t = t + 1.0f;
    t[0] = __builtin_amdgcn_rcpf(t[0]); t[1] = __builtin_amdgcn_rcpf(t[1]);
    return u * t;
}
// one dword of two bf16 -> act(x * scale + shift) -> one dword of two bf16 (round to nearest even: v_cvt_pk_bf16_f32)
template <bool SILU>
__device__ __forceinline__ unsigned act_pair_bf16(unsigned w, f32x2 sc, f32x2 sh) {
    f32x2 x;
    x[0] = __uint_as_float(w << 16); x[1] = __uint_as_float(w & 0xffff0000u);
    f32x2 u = x * sc + sh;
    if constexpr (SILU) u = silu2_f(u);
    const bf16x2 o = __builtin_convertvector(u, bf16x2);
    return *reinterpret_cast<const unsigned*>(&o);
}
// two bf16 of one dword -> two fp32 (v_lshlrev / v_and), and the packed silu'(u): 5 v_pk_* + 2 v_exp + 2 v_rcp for two elements
__device__ __forceinline__ f32x2 bf16pair_f32(unsigned w) {
    f32x2 x;
    x[0] = __uint_as_float(w << 16); x[1] = __uint_as_float(w & 0xffff0000u);
    return x;
}
__device__ __forceinline__ unsigned f32pair_bf16(f32x2 v) {
    const bf16x2 o = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<const unsigned*>(&o);
}
__device__ __forceinline__ f32x2 dsilu2_f(f32x2 u) {
    f32x2 t = u * -1.4426950408889634f;
    t[0] = __builtin_amdgcn_exp2f(t[0]); t[1] = __builtin_amdgcn_exp2f(t[1]);
    t = t + 1.0f;
    f32x2 s;
    s[0] = __builtin_amdgcn_rcpf(t[0]); s[1] = __builtin_amdgcn_rcpf(t[1]);
    return s * (u * (1.0f - s) + 1.0f);
}
// d silu(u)/du = s*(1+u*(1-s)),  s = sigmoid(u)
__device__ __forceinline__ float dsilu_f(float u) { float s = __builtin_amdgcn_rcpf(1.0f + __expf(-u)); return s * (1.0f + u * (1.0f - s)); }

// ---- MFMA: one 32x32 tile, K-chunk of 16 (8 elements per lane: k = 8*(lane>>5)+j) ---
// D[row=(r&3)+8*(r>>2)+4*(lane>>5)][col=lane&31] += sum_k A[row][k]*B[k][col]
// bf16: one v_mfma_f32_32x32x16_bf16.  f32: eight v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain);
// MFMA j pairs lane-group g's element j of A with the same of B, so any (g,j)->k map is valid
// as long as A and B are loaded with the same one.
__device__ __forceinline__ void mma16(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x16& acc, const f32x8& a, const f32x8& b) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
}
// accumulator row owned by (lane, reg)
__device__ __forceinline__ int acc_row(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
