// box_convert and box_iou: the two elementwise torchvision.ops primitives the
// reference's model files call (src/models/ssd300.py:249-251 and friends).
// Arithmetic is written in the exact operation order of the published
// torchvision formulas; the build uses -ffp-contract=off so nothing is fused.
#include "detbox_internal.h"

using namespace detbox;

namespace {

__device__ inline float4 to_xyxy(float4 b, int fmt) {
    if (fmt == DETBOX_FMT_XYWH) return make_float4(b.x, b.y, b.x + b.z, b.y + b.w);
    if (fmt == DETBOX_FMT_CXCYWH)
        return make_float4(b.x - 0.5f * b.z, b.y - 0.5f * b.w, b.x + 0.5f * b.z, b.y + 0.5f * b.w);
    return b;
}

__device__ inline float4 from_xyxy(float4 b, int fmt) {
    if (fmt == DETBOX_FMT_XYWH) return make_float4(b.x, b.y, b.z - b.x, b.w - b.y);
    if (fmt == DETBOX_FMT_CXCYWH)
        return make_float4((b.x + b.z) / 2.0f, (b.y + b.w) / 2.0f, b.z - b.x, b.w - b.y);
    return b;
}

__global__ __launch_bounds__(256) void box_convert_kernel(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                          int64_t n, int in_fmt, int out_fmt) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 b = in[i];
    if (in_fmt != out_fmt) b = from_xyxy(to_xyxy(b, in_fmt), out_fmt);
    out[i] = b;
}

// One thread per (row of boxes1, 4 consecutive columns of boxes2); boxes2 tile staged in LDS.
constexpr int IOU_TILE_M = 256;

__global__ __launch_bounds__(256) void box_iou_kernel(const float4 *__restrict__ b1, int64_t n,
                                                      const float4 *__restrict__ b2, int64_t m,
                                                      float *__restrict__ out) {
    __shared__ float4 s_box[IOU_TILE_M];
    __shared__ float s_area[IOU_TILE_M];
    const int64_t m0 = (int64_t)blockIdx.y * IOU_TILE_M;
    const int mt = (int)min((int64_t)IOU_TILE_M, m - m0);
    for (int j = threadIdx.x; j < mt; j += blockDim.x) {
        float4 g = b2[m0 + j];
        s_box[j] = g;
        s_area[j] = (g.z - g.x) * (g.w - g.y);
    }
    __syncthreads();
    // 256 threads = 4 rows x 64 column lanes; each block covers 64 rows
    const int lane = threadIdx.x & 63;
    const int rsub = threadIdx.x >> 6;
    for (int r = rsub; r < 64; r += 4) {
        int64_t i = (int64_t)blockIdx.x * 64 + r;
        if (i >= n) break;
        float4 a = b1[i];
        float area_a = (a.z - a.x) * (a.w - a.y);
        for (int j = lane; j < mt; j += 64) {
            float4 g = s_box[j];
            float ltx = fmaxf(a.x, g.x), lty = fmaxf(a.y, g.y);
            float rbx = fminf(a.z, g.z), rby = fminf(a.w, g.w);
            float w = fmaxf(rbx - ltx, 0.0f), h = fmaxf(rby - lty, 0.0f);
            float inter = w * h;
            float uni = (area_a + s_area[j]) - inter;
            out[i * m + m0 + j] = inter / uni;
        }
    }
}

}  // namespace

extern "C" int detbox_box_convert(const float *boxes, float *out, int64_t n, int in_fmt, int out_fmt, void *stream) {
    DETBOX_CHECK_ARG(n >= 0, "negative n");
    DETBOX_CHECK_ARG(in_fmt >= 0 && in_fmt <= 2 && out_fmt >= 0 && out_fmt <= 2, "unknown box format");
    if (n == 0) return DETBOX_OK;
    DETBOX_CHECK_ARG(boxes && out, "null pointer");
    DETBOX_CHECK_ARG(((uintptr_t)boxes % 16 == 0) && ((uintptr_t)out % 16 == 0), "boxes must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope ps("box_convert", s);
    box_convert_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(
        reinterpret_cast<const float4 *>(boxes), reinterpret_cast<float4 *>(out), n, in_fmt, out_fmt);
    DETBOX_LAUNCH_CHECK();
    return DETBOX_OK;
}

extern "C" int detbox_box_iou(const float *boxes1, int64_t n, const float *boxes2, int64_t m, float *iou, void *stream) {
    DETBOX_CHECK_ARG(n >= 0 && m >= 0, "negative size");
    if (n == 0 || m == 0) return DETBOX_OK;
    DETBOX_CHECK_ARG(boxes1 && boxes2 && iou, "null pointer");
    DETBOX_CHECK_ARG(((uintptr_t)boxes1 % 16 == 0) && ((uintptr_t)boxes2 % 16 == 0), "boxes must be 16-byte aligned");
    int64_t gx = (n + 63) / 64, gy = (m + IOU_TILE_M - 1) / IOU_TILE_M;
    DETBOX_CHECK_ARG(gy <= 65535, "boxes2 too long (m > 16M)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope ps("box_iou", s);
    box_iou_kernel<<<dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, s>>>(
        reinterpret_cast<const float4 *>(boxes1), n, reinterpret_cast<const float4 *>(boxes2), m, iou);
    DETBOX_LAUNCH_CHECK();
    return DETBOX_OK;
}
