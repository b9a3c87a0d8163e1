// common.cuh — shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wave = 64 lanes everywhere in this tree; nothing here is meant to compile for another target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LEANN_EMPTY 0xFFFFFFFFu
#define WAVE 64

// ---------------------------------------------------------------------------------------------
// Orderable keys.  key64 = orderable(dist) << 32 | id << 1 | expanded_flag.  A single u64 compare
// implements the total order (dist, id) used by every beam / merge decision, so CPU and GPU make
// identical choices (ids are < 2^31 per shard; the flag never decides because ids are unique).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f32_orderable(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float orderable_f32(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t make_key(float dist, uint32_t id) {
    return ((uint64_t)f32_orderable(dist) << 32) | ((uint64_t)id << 1);
}
__device__ __forceinline__ uint32_t key_id(uint64_t k) { return (uint32_t)(k & 0xFFFFFFFFull) >> 1; }
__device__ __forceinline__ float key_dist(uint64_t k) { return orderable_f32((uint32_t)(k >> 32)); }

// ---------------------------------------------------------------------------------------------
// Hashing shared with oracle/oracle.c (orc_mix64 / orc_hash3 / orc_gauss / orc_level).
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ uint64_t hash3(uint64_t seed, uint64_t a, uint64_t b) {
    return mix64(mix64(seed ^ (a * 0xD1342543DE82EF95ull)) ^ (b * 0xA24BAED4963EE407ull));
}
__host__ __device__ __forceinline__ float gauss_ih4(uint64_t seed, uint64_t a, uint64_t b) {
    uint64_t h = hash3(seed, a, b);
    int32_t s = (int32_t)((h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48));
    return (float)(s - 131070) * 2.6428996e-05f;
}
__host__ __device__ __forceinline__ uint32_t node_level(uint64_t seed, uint64_t i, uint32_t M) {
    uint64_t u = hash3(seed, i, 0x4C45564Cull);
    uint64_t thr = 0xFFFFFFFFFFFFFFFFull;
    uint32_t l = 0;
    while (l < 15) {
        thr /= M;
        if (u >= thr) break;
        l++;
    }
    return l;
}

// ---------------------------------------------------------------------------------------------
// Canonical ("wave order") dot product — DESIGN.md §3, oracle/oracle.c:orc_dot_canon.
// Lane l owns the 4 strided accumulators a[4l..4l+3]; chunk t covers elements 256t .. 256t+255.
// After the fmaf chains: (a0+a1)+(a2+a3) in-lane, then the perfect adjacent-pair tree over the 64 lane sums (levels 1, 2, 4, 8, 16,
// 32) — the same pairs, hence the same bits, as an xor butterfly, but on the DPP path of the VALU: quad permutes for levels 1 and 2,
// row shifts for 4 and 8 (the pair sum lands in the upper lane of each pair), row_bcast15 / row_bcast31 across rows; lane 63 ends
// with the total, which v_readlane broadcasts.  (__shfl_xor compiles to ds_bpermute: six dependent LDS round trips per row batch.)
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_pair_add(float v) { // lanes without a source, or outside ROW_MASK, add 0
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_tree_sum(float s) {
    s = dpp_pair_add<0xB1, 0xf>(s);  // quad_perm [1,0,3,2]: lane ^ 1
    s = dpp_pair_add<0x4E, 0xf>(s);  // quad_perm [2,3,0,1]: lane ^ 2
    s = dpp_pair_add<0x114, 0xf>(s); // row_shr:4  -> lanes 4..7, 12..15 of each row hold the sums of 8
    s = dpp_pair_add<0x118, 0xf>(s); // row_shr:8  -> lanes 12..15: the row's 16
    s = dpp_pair_add<0x142, 0xa>(s); // row_bcast:15 into rows 1 and 3 -> lanes 28..31, 60..63: 32
    s = dpp_pair_add<0x143, 0xc>(s); // row_bcast:31 into rows 2 and 3 -> lanes 60..63: all 64
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 63));
}
// minimum over the wave on the same DPP ladder (lanes without a source keep their own value); every lane gets the result
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_pair_min(uint32_t v) {
    return min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = dpp_pair_min<0xB1, 0xf>(v);
    v = dpp_pair_min<0x4E, 0xf>(v);
    v = dpp_pair_min<0x114, 0xf>(v);
    v = dpp_pair_min<0x118, 0xf>(v);
    v = dpp_pair_min<0x142, 0xa>(v);
    v = dpp_pair_min<0x143, 0xc>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float lane4_sum(float4 a) { return (a.x + a.y) + (a.z + a.w); }

__device__ __forceinline__ void fma4(float4 &acc, const float4 &a, const float4 &b) {
    acc.x = fmaf(a.x, b.x, acc.x);
    acc.y = fmaf(a.y, b.y, acc.y);
    acc.z = fmaf(a.z, b.z, acc.z);
    acc.w = fmaf(a.w, b.w, acc.w);
}

// Load lane's float4 of chunk t from a 16-byte aligned, zero-padded row of leading dimension ld.
__device__ __forceinline__ float4 row_load4(const float *__restrict__ row, uint32_t ld, int t, int lane) {
    uint32_t j = 256u * t + 4u * lane;
#ifdef LEANN_NT_ROWS
    if (j < ld) { // experiment (scripts/variant.sh): streamed rows bypass-ish the caches
        typedef float vf4 __attribute__((ext_vector_type(4)));
        vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(row + j));
        return make_float4(v.x, v.y, v.z, v.w);
    }
#else
    if (j < ld) return *reinterpret_cast<const float4 *>(row + j);
#endif
    return make_float4(0.f, 0.f, 0.f, 0.f);
}
// Guarded scalar loads (query rows may be unaligned / unpadded: leading dimension = d).
__device__ __forceinline__ float4 vec_load4_guard(const float *__restrict__ v, uint32_t d, int t, int lane) {
    uint32_t j = 256u * t + 4u * lane;
    float4 r;
    r.x = j + 0 < d ? v[j + 0] : 0.f;
    r.y = j + 1 < d ? v[j + 1] : 0.f;
    r.z = j + 2 < d ? v[j + 2] : 0.f;
    r.w = j + 3 < d ? v[j + 3] : 0.f;
    return r;
}

// ---------------------------------------------------------------------------------------------
// Block-wide bitonic sort of n (power of two, <= SEG) u64 keys in LDS, ascending (scan.hip / recompute.hip top-k).
// ---------------------------------------------------------------------------------------------
#define SEG 2048
__device__ __forceinline__ void bitonic_sort_lds(uint64_t *k, int n /* power of two */) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < n / 2; i += blockDim.x) {
                int lo = 2 * i - (i & (stride - 1));
                int hi = lo + stride;
                bool up = ((lo & size) == 0);
                uint64_t a = k[lo], b = k[hi];
                if ((a > b) == up) { k[lo] = b; k[hi] = a; }
            }
        }
    }
    __syncthreads();
}

#define HIP_CHECK_RET(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            leann_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                            __LINE__);                                                       \
            return LEANN_ERR_DEVICE;                                                         \
        }                                                                                    \
    } while (0)

extern "C" void leann_set_error(const char *fmt, ...);
