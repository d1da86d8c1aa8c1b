// Internal helpers shared by the detbox translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/detbox.h"

namespace detbox {

// thread-local error text returned by detbox_last_error()
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);

// per-kernel hipEvent timing (detbox_profile_*); begin/end are no-ops while disabled
void prof_begin(const char *name, hipStream_t s);
void prof_end(const char *name, hipStream_t s);

struct ProfScope {
    const char *name; hipStream_t s;
    ProfScope(const char *n, hipStream_t st) : name(n), s(st) { prof_begin(name, s); }
    ~ProfScope() { prof_end(name, s); }
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Carves 256-byte aligned regions out of a caller-provided workspace.
struct Carver {
    char *base; size_t off;
    explicit Carver(void *p) : base(static_cast<char *>(p)), off(0) {}
    template <typename T> T *take(size_t count) {
        T *r = reinterpret_cast<T *>(base + off);
        off = align_up(off + count * sizeof(T), 256);
        return r;
    }
};

// ---- order step of the batched NMS + result rows (merge.hip), driven from nms.hip -----------------------------
// How the kept runs of one image are cut into tiles: every SP-th run element is a sample, every Q-th merged sample a
// splitter, at most T_max tiles per image; an image of at most one_tile_rows kept rows is not cut.  ok == false: the shape is outside what the tile kernels hold
// (more than 256 segments per image, or tiles that would not fit) and the caller takes the level-by-level merge.
struct MergePlan { bool ok; int SP, Q, T_max, one_tile_rows; };
MergePlan merge_plan(int64_t B, int64_t stride, int64_t S);
size_t merge_workspace_bytes(const MergePlan &p, int64_t B, int64_t S);

// Where the result rows go (all device pointers; wire may be null).
struct RowsOut {
    const float4 *boxes; const float *scores; const int64_t *class_ids;
    const int64_t *image_ids; const int32_t *image_sizes;
    int64_t *rows_ids; double *rows_vals; int32_t *row_offsets;
    int hdr_rows; uint32_t *wire;
};
// kept: per (image, segment) run of (ordered score key, row) pairs starting at the segment's first row; kcount [B,S].
// Writes keep_counts, keep (unless null) and, with rows != null, the result rows.
int merge_rows_launch(const MergePlan &p, const float2 *kept, const int32_t *kcount, const int32_t *seg_offsets,
                      int64_t B, int64_t stride, int64_t S, int64_t *keep, int32_t *keep_counts, const RowsOut *rows,
                      void *workspace, hipStream_t st);

#define DETBOX_CHECK_ARG(cond, msg)                                 \
    do { if (!(cond)) { detbox::set_error("%s: %s", __func__, msg);   \
                        return DETBOX_ERR_INVALID; } } while (0)

#define DETBOX_HIP(call)                                              \
    do { hipError_t e_ = (call);                                      \
         if (e_ != hipSuccess) return detbox::hip_fail(e_, #call); } while (0)

#define DETBOX_LAUNCH_CHECK()                                         \
    do { hipError_t e_ = hipGetLastError();                           \
         if (e_ != hipSuccess) return detbox::hip_fail(e_, "kernel launch"); } while (0)

// ---- device helpers ---------------------------------------------------------------------

// Monotone map of fp32 onto uint32: a < b  <=>  okey(a) < okey(b) (for non-NaN, with -0 == +0).
__device__ __host__ inline uint32_t okey_from_bits(uint32_t u) {
    if (u == 0x80000000u) u = 0u;                       // -0 -> +0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __host__ inline uint32_t bits_from_okey(uint32_t k) {
    return (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
}

// Workgroups are dealt to the eight XCDs round-robin by linear workgroup id, and every XCD has an L2 of
// its own.  For a grid (tiles, images) this returns the (tile, image) this workgroup should work on so that
// all the workgroups of one image run on one XCD and that image's rows are fetched into one L2 only.
// A bijection of the grid onto itself; the identity when the number of images is not a multiple of 8.
struct TileOfImage { int x, y; };
__device__ inline TileOfImage xcd_affine_tile() {
    TileOfImage t;
    t.x = (int)blockIdx.x; t.y = (int)blockIdx.y;
    const unsigned T = gridDim.x, B = gridDim.y;
    if ((B & 7u) == 0u) {
        const unsigned L = blockIdx.x + T * blockIdx.y;
        const unsigned idx = L >> 3;
        t.y = (int)((L & 7u) + 8u * (idx / T));
        t.x = (int)(idx % T);
    }
    return t;
}

// Stores of outputs that nobody on the GPU reads back soon (result rows, training targets, gradients): non-temporal,
// so that a stream of write-once data does not displace reusable lines from the L2 and the memory-side cache.
typedef float detbox_v4f __attribute__((ext_vector_type(4)));
__device__ inline void store_nt(float4 *p, const float4 v) {
    detbox_v4f x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<detbox_v4f *>(p));
}
__device__ inline void store_nt(float *p, float v) { __builtin_nontemporal_store(v, p); }
__device__ inline void store_nt(double *p, double v) { __builtin_nontemporal_store(v, p); }
__device__ inline void store_nt(int64_t *p, int64_t v) { __builtin_nontemporal_store(v, p); }
__device__ inline void store_nt(uint32_t *p, uint32_t v) { __builtin_nontemporal_store(v, p); }

// Inclusive prefix sum over the 64 lanes with data-parallel-primitive moves: shifts inside the rows of 16, then the row
// totals are broadcast into the following rows.  Every lane of the wave must be active.
__device__ inline uint32_t wave_inclusive_add(uint32_t v) {
#define DETBOX_DPP_ADD(ctrl, rows) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, false)
    DETBOX_DPP_ADD(0x111, 0xf);     // row_shr:1
    DETBOX_DPP_ADD(0x112, 0xf);     // row_shr:2
    DETBOX_DPP_ADD(0x114, 0xf);     // row_shr:4
    DETBOX_DPP_ADD(0x118, 0xf);     // row_shr:8
    DETBOX_DPP_ADD(0x142, 0xa);     // row_bcast:15 into rows 1 and 3
    DETBOX_DPP_ADD(0x143, 0xc);     // row_bcast:31 into rows 2 and 3
#undef DETBOX_DPP_ADD
    return v;
}

// Value of lane (lane ^ J) for every lane.  Distances inside a row of 16 lanes are data-parallel-primitive moves in the
// VALU (1, 2: quad permutations; 4: the two shifts by four, each kept in the quads it is right for; 8: a rotation of the
// row by eight); 16 and 32 cross rows and go through the LDS crossbar.
template <int J>
__device__ inline uint32_t lane_xor_u32(const uint32_t v) {
    if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);      // quad_perm:[1,0,3,2]
    else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true); // quad_perm:[2,3,0,1]
    else if constexpr (J == 4) {
        const int up = __builtin_amdgcn_mov_dpp((int)v, 0x104, 0xf, 0x5, true);                             // row_shl:4 into quads 0 and 2
        return (uint32_t)__builtin_amdgcn_update_dpp(up, (int)v, 0x114, 0xf, 0xa, false);                       // row_shr:4 into quads 1 and 3
    } else if constexpr (J == 8) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xf, 0xf, true); // row_ror:8
    else return (uint32_t)__shfl_xor((int)v, J);
}

template <int J>
__device__ inline unsigned long long lane_xor_u64(const unsigned long long v) {
    return ((unsigned long long)lane_xor_u32<J>((uint32_t)(v >> 32)) << 32) | lane_xor_u32<J>((uint32_t)v);
}

template <int J>
__device__ inline float lane_xor_f32(const float v) { return __uint_as_float(lane_xor_u32<J>(__float_as_uint(v))); }

}  // namespace detbox
