// Shared device/host helpers for the gfx950 DDPM kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// Global memory through a workgroup-uniform base (scalar registers) plus a 32-bit byte offset per lane: one address
// register per access instead of a 64-bit pair, and no 64-bit vector arithmetic to form it.
template <class T>
__device__ __forceinline__ T ld_at(const void* base, unsigned byte_off) { return *(const T*)((const char*)base + byte_off); }
template <class T>
__device__ __forceinline__ void st_at(void* base, unsigned byte_off, const T& v) { *(T*)((char*)base + byte_off) = v; }

#define DDPM_OK 0
#define DDPM_EINVAL (-1)
#define DDPM_EHIP (-2)

// Activation codes shared by every kernel and by the host side.
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2, ACT_RELU = 3, ACT_TANH = 4, ACT_SIGMOID = 5, ACT_HARDSIGMOID = 6, ACT_RELU6 = 7,
       ACT_LEAKY_RELU = 8, ACT_SELU = 9,
       // the two with a learnable scalar b (neural_nets/activations.py:25-63 of the reference): z * sigmoid(b z), z * hardsigmoid(b z)
       ACT_SWISH = 10, ACT_HARDSWISH = 11, ACT_LAST = 11 };
#define ACT_HAS_BETA(act) ((act) == ACT_SWISH || (act) == ACT_HARDSWISH)
#define SELU_SCALE 1.0507009873554805f
#define SELU_ALPHA 1.6732632423543772f
#define LEAKY_SLOPE 0.05f   /* the reference registry's leaky_relu slope (neural_nets/activations.py:110) */

void ddpm_set_error(const char* fmt, ...);
int ddpm_check_launch(const char* what);

#define DDPM_REQUIRE(cond, ...)                  \
    do {                                         \
        if (!(cond)) {                           \
            ddpm_set_error(__VA_ARGS__);         \
            return DDPM_EINVAL;                  \
        }                                        \
    } while (0)

// erf via Abramowitz & Stegun 7.1.26 (|abs error| < 1.5e-7, i.e. fp32-level): one exp, one reciprocal, a 5-term
// polynomial.  The exact-erf GELU (torch GELU(approximate="none"), the reference's default activation) is evaluated
// for every element of ~80 tensors per pass, so its instruction count matters; and since erf(z/sqrt2) needs
// exp(-z^2/2), the Gaussian factor of GELU's derivative comes for free.
__device__ __forceinline__ void gelu_parts(float z, float& cdf, float& gauss) {
    // constants folded onto |z| itself: 0.3275911 / sqrt2, sqrt(log2(e) / 2), and half of the five coefficients
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.23164189f * az);        // v_rcp_f32 (1 ulp); an IEEE divide costs 11 instructions
    const float q = az * 0.84932180f;
    gauss = __builtin_amdgcn_exp2f(-(q * q));                              // exp(-z^2/2) as one v_exp_f32
    const float half_poly = t * (0.127414796f + t * (-0.142248368f + t * (0.7107068705f + t * (-0.7265760135f + t * 0.5307027145f))));
    const float e2 = half_poly * gauss;                                    // (1 - erf|z/sqrt2|) / 2 = the smaller tail of Phi
    cdf = z < 0.f ? e2 : 1.0f - e2;
}

// gelu(z) = z * Phi(z) with the same erf, rearranged for the forward pass: with e = poly * gauss (= 1 - erf|z/sqrt2|),
// z * Phi(z) = max(z, 0) - |z| * e / 2 for either sign of z.  No sign select, no 1 - e (whose cancellation costs the negative tail
// its relative accuracy), two instructions fewer per element than forming Phi first; the halves sit in the coefficients.
__device__ __forceinline__ float gelu_fwd(float z) {
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.23164189f * az);
    const float q = az * 0.84932180f;
    const float gauss = __builtin_amdgcn_exp2f(-(q * q));
    const float half_poly = t * (0.127414796f + t * (-0.142248368f + t * (0.7107068705f + t * (-0.7265760135f + t * 0.5307027145f))));
    return fmaxf(z, 0.f) - (az * gauss) * half_poly;
}

// logistic function on the fast hardware exp2 / reciprocal (1-ulp) instructions
__device__ __forceinline__ float sigmoid_fast(float z) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
}

__device__ __forceinline__ float act_fwd(float z, int act) {
    switch (act) {
        case ACT_GELU: return gelu_fwd(z);
        case ACT_SILU: return z * sigmoid_fast(z);
        case ACT_RELU: return z > 0.f ? z : 0.f;
        case ACT_TANH: return tanhf(z);
        case ACT_SIGMOID: return sigmoid_fast(z);
        case ACT_HARDSIGMOID: return fminf(fmaxf(z + 3.f, 0.f), 6.f) * (1.f / 6.f);
        case ACT_RELU6: return fminf(fmaxf(z, 0.f), 6.f);
        case ACT_LEAKY_RELU: return z > 0.f ? z : LEAKY_SLOPE * z;
        case ACT_SELU: return SELU_SCALE * (z > 0.f ? z : SELU_ALPHA * (__expf(z) - 1.f));
        default: return z;
    }
}

// d act(z) / dz
__device__ __forceinline__ float act_bwd(float z, int act) {
    switch (act) {
        case ACT_GELU: { float c, g; gelu_parts(z, c, g); return c + z * 0.39894228040143268f * g; }
        case ACT_SILU: {
            const float s = sigmoid_fast(z);
            return s * (1.0f + z * (1.0f - s));
        }
        case ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case ACT_TANH: {
            const float t = tanhf(z);
            return 1.0f - t * t;
        }
        case ACT_SIGMOID: {
            const float s = sigmoid_fast(z);
            return s * (1.0f - s);
        }
        case ACT_HARDSIGMOID: return (z > -3.f && z < 3.f) ? (1.f / 6.f) : 0.f;
        case ACT_RELU6: return (z > 0.f && z < 6.f) ? 1.f : 0.f;
        case ACT_LEAKY_RELU: return z > 0.f ? 1.f : LEAKY_SLOPE;
        case ACT_SELU: return z > 0.f ? SELU_SCALE : SELU_SCALE * SELU_ALPHA * __expf(z);
        default: return 1.f;
    }
}

// The same with the activation's learnable scalar b (ignored by the parameter-free codes).
__device__ __forceinline__ float hardsigmoid_of(float u) { return fminf(fmaxf(u + 3.f, 0.f), 6.f) * (1.f / 6.f); }

__device__ __forceinline__ float act_fwd(float z, int act, float b) {
    if (act == ACT_SWISH) return z * sigmoid_fast(b * z);
    if (act == ACT_HARDSWISH) return z * hardsigmoid_of(b * z);
    return act_fwd(z, act);
}

__device__ __forceinline__ float act_bwd(float z, int act, float b) {
    if (act == ACT_SWISH) {
        const float s = sigmoid_fast(b * z);
        return s * (1.0f + b * z * (1.0f - s));
    }
    if (act == ACT_HARDSWISH) {
        const float u = b * z;
        return hardsigmoid_of(u) + ((u > -3.f && u < 3.f) ? u * (1.f / 6.f) : 0.f);
    }
    return act_bwd(z, act);
}

// d act(z; b) / db
__device__ __forceinline__ float act_dbeta(float z, int act, float b) {
    if (act == ACT_SWISH) {
        const float s = sigmoid_fast(b * z);
        return z * z * s * (1.0f - s);
    }
    if (act == ACT_HARDSWISH) {
        const float u = b * z;
        return (u > -3.f && u < 3.f) ? z * z * (1.f / 6.f) : 0.f;
    }
    return 0.f;
}

// Runs f(integral_constant<int, A>) with A = the activation code when it is one of the hot ones (gelu, silu, none), so that
// act_fwd / act_bwd inside f fold to straight-line code; every other code runs f with A = -1, where the callee keeps the
// runtime switch.  A loop over elements that calls act_fwd(x, act) with a runtime code expands the whole switch (tanhf and
// expf bodies included) around every element: several hundred instructions each and an instruction stream that no longer
// fits the instruction cache.
template <int A> struct ActTag { static constexpr int value = A; };
template <typename F>
__device__ __forceinline__ void act_dispatch(int act, F&& f) {
    if (act == ACT_GELU) f(ActTag<ACT_GELU>{});
    else if (act == ACT_SILU) f(ActTag<ACT_SILU>{});
    else if (act == ACT_NONE) f(ActTag<ACT_NONE>{});
    else f(ActTag<-1>{});
}
#define ACT_OF(tag, runtime) (decltype(tag)::value >= 0 ? decltype(tag)::value : (runtime))

// ---------------------------------------------------------------- Philox4x32-10 (counter-based RNG)
struct Philox4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

// Keep-mask bits for the 8 consecutive elements whose first flat index is 8*vec_index, at a dropout
// site.  Bit j set => element j kept.  thresh = keep_prob * 65536.  Depends only on (seed, site, vec_index),
// so the forward, data-gradient and weight-gradient kernels regenerate identical masks.
__device__ __forceinline__ uint32_t dropout_keep8(uint64_t seed, uint32_t site, uint64_t vec_index, uint32_t thresh) {
    const Philox4 r = philox4x32_10((uint32_t)vec_index, (uint32_t)(vec_index >> 32), site, 0x0d0b0a17u,
                                    (uint32_t)seed, (uint32_t)(seed >> 32));
    uint32_t m = 0;
    m |= ((r.x & 0xFFFFu) < thresh) << 0;
    m |= ((r.x >> 16) < thresh) << 1;
    m |= ((r.y & 0xFFFFu) < thresh) << 2;
    m |= ((r.y >> 16) < thresh) << 3;
    m |= ((r.z & 0xFFFFu) < thresh) << 4;
    m |= ((r.z >> 16) < thresh) << 5;
    m |= ((r.w & 0xFFFFu) < thresh) << 6;
    m |= ((r.w >> 16) < thresh) << 7;
    return m;
}

// Four N(0,1) samples for flat element indices 4*q .. 4*q+3 of stream `stream_id`.
__device__ __forceinline__ f32x4 philox_normal4(uint64_t seed, uint32_t stream_id, uint64_t q) {
    const Philox4 r = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), stream_id, 0x6a09e667u, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u0 = ((float)r.x + 1.0f) * 2.3283064365386963e-10f;  // (0,1]
    const float u1 = (float)r.y * 2.3283064365386963e-10f;
    const float u2 = ((float)r.z + 1.0f) * 2.3283064365386963e-10f;
    const float u3 = (float)r.w * 2.3283064365386963e-10f;
    const float ra = sqrtf(-2.0f * __logf(u0)), rb = sqrtf(-2.0f * __logf(u2));
    float s0, c0, s1, c1;
    __sincosf(6.283185307179586f * u1, &s0, &c0);
    __sincosf(6.283185307179586f * u3, &s1, &c1);
    f32x4 o;
    o[0] = ra * c0; o[1] = ra * s0; o[2] = rb * c1; o[3] = rb * s1;
    return o;
}

// Sums for GroupNorm statistics straight from packed bf16 (v_dot2c_f32_bf16: both products exact, accumulated in fp32), one
// instruction per channel pair and sum instead of unpack + add / fma per channel: s[q] += v[2q] + v[2q+1] and
// ss[q] += v[2q]^2 + v[2q+1]^2.  Channel pairs never straddle a normalisation group where this is used (groups of 2, 4, 8 ...).
__device__ __forceinline__ void pair_sums(const bf16x8& v, float (&s)[4], float (&ss)[4]) {
    const bf16x2 one = {(bf16)1.f, (bf16)1.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bf16x2 p = {v[2 * q], v[2 * q + 1]};
        s[q] = __builtin_amdgcn_fdot2_f32_bf16(p, one, s[q], false);
        ss[q] = __builtin_amdgcn_fdot2_f32_bf16(p, p, ss[q], false);
    }
}
// the same into the even entries of per-channel arrays (the odd entries stay as they are: zero)
__device__ __forceinline__ void pair_sums(const bf16x8& v, float (&s)[8], float (&ss)[8]) {
    const bf16x2 one = {(bf16)1.f, (bf16)1.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bf16x2 p = {v[2 * q], v[2 * q + 1]};
        s[2 * q] = __builtin_amdgcn_fdot2_f32_bf16(p, one, s[2 * q], false);
        ss[2 * q] = __builtin_amdgcn_fdot2_f32_bf16(p, p, ss[2 * q], false);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Column sums across a workgroup: thread tid owns 8 per-channel values of vector column (tid % vps), vps a power of two
// <= 8; on return every thread holds the sum over all threads of its column.  s_red: LDS, (THREADS/64)*64 floats (>= 128).
template <int THREADS>
__device__ __forceinline__ void block_reduce_cols(float (&v)[8], int vps, float* s_red /* [THREADS/64][64] */, int tid) {
    // sum v[] over all threads with the same (tid % vps); result valid in every thread afterwards.
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float a = v[j];
        for (int o = 32; o >= vps; o >>= 1) a += __shfl_xor(a, o, 64);
        v[j] = a;
    }
    if constexpr (THREADS > 64) {
        const int wave = tid >> 6, lane = tid & 63;
        __syncthreads();
        if (lane < vps) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s_red[wave * 64 + lane * 8 + j] = v[j];
        }
        __syncthreads();
        // waves added in ascending order (a fixed order: the sums are reproducible), two 16-byte reads per wave and only a
        // few waves' reads in flight: fully unrolled scalar reads hold THREADS / 8 registers at once
        const int cv = lane % vps;
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int w = 0; w < THREADS / 64; ++w) {
            lo += *(const f32x4*)(s_red + w * 64 + cv * 8);
            hi += *(const f32x4*)(s_red + w * 64 + cv * 8 + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
    }
}
