// Localisation loss of the SSD and RetinaNet heads, fused with its mask:
//   F.smooth_l1_loss(out_locs[pos_mask], target_locs[pos_mask], reduction='sum'),  pos_mask = target_labels > 0
// (src/models/ssd300.py:283, src/models/retinanet.py:211).  The reference builds the boolean-mask gathers
// (a nonzero() with a host synchronisation, two [N_pos, 4] copies) and then reduces; here one pass reads
// the labels, touches the two box rows of the positive anchors only and accumulates in float64.
// smooth_l1 with beta = 1 (the default the reference uses): 0.5 * d^2 for |d| < 1, |d| - 0.5 otherwise;
// derivative d for |d| < 1, sign(d) otherwise.
#include "detbox_internal.h"

using namespace detbox;

namespace {

constexpr int LOSS_BLOCK = 256;

__device__ inline float smooth_l1(float x, float y) {
    const float z = fabsf(x - y);
    return z < 1.0f ? (0.5f * z) * z : z - 0.5f;
}

__device__ inline float smooth_l1_grad(float x, float y) {
    const float d = x - y;
    const float z = fabsf(d);
    return z < 1.0f ? d : (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));
}

__global__ __launch_bounds__(LOSS_BLOCK) void smooth_l1_forward_kernel(const float4 *__restrict__ input,
                                                                       const float4 *__restrict__ target,
                                                                       const int64_t *__restrict__ labels, long long rows,
                                                                       double *__restrict__ loss_sum) {
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * LOSS_BLOCK + threadIdx.x; i < rows; i += (long long)gridDim.x * LOSS_BLOCK) {
        if (labels[i] > 0) {
            const float4 a = input[i], t = target[i];
            acc += (double)smooth_l1(a.x, t.x);
            acc += (double)smooth_l1(a.y, t.y);
            acc += (double)smooth_l1(a.z, t.z);
            acc += (double)smooth_l1(a.w, t.w);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ double s_part[LOSS_BLOCK / 64];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < LOSS_BLOCK / 64; ++w) t += s_part[w];
        if (t != 0.0) atomicAdd(loss_sum, t);
    }
}

__global__ __launch_bounds__(LOSS_BLOCK) void smooth_l1_backward_kernel(const float4 *__restrict__ input,
                                                                        const float4 *__restrict__ target,
                                                                        const int64_t *__restrict__ labels, long long rows,
                                                                        const float *__restrict__ grad_scale,
                                                                        float4 *__restrict__ grad) {
    const float g = *grad_scale;
    for (long long i = (long long)blockIdx.x * LOSS_BLOCK + threadIdx.x; i < rows; i += (long long)gridDim.x * LOSS_BLOCK) {
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (labels[i] > 0) {
            const float4 a = input[i], t = target[i];
            r = make_float4(g * smooth_l1_grad(a.x, t.x), g * smooth_l1_grad(a.y, t.y), g * smooth_l1_grad(a.z, t.z),
                            g * smooth_l1_grad(a.w, t.w));
        }
        grad[i] = r;
    }
}

int check_rows(const float *input, const float *target, const int64_t *labels, int64_t rows) {
    DETBOX_CHECK_ARG(rows >= 0 && rows < (1ll << 40), "bad size");
    DETBOX_CHECK_ARG(rows == 0 || (input && target && labels), "null pointer");
    DETBOX_CHECK_ARG((uintptr_t)input % 16 == 0 && (uintptr_t)target % 16 == 0, "box rows must be 16-byte aligned");
    return DETBOX_OK;
}

unsigned loss_grid(int64_t rows) {
    int64_t g = (rows + LOSS_BLOCK - 1) / LOSS_BLOCK;
    if (g > 4096) g = 4096;                 // a few workgroups per CU; the rest of the rows by the grid stride
    return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int detbox_smooth_l1_masked_forward(const float *input, const float *target, const int64_t *labels,
                                               int64_t rows, double *loss_sum, void *stream) {
    if (int rc = check_rows(input, target, labels, rows)) return rc;
    DETBOX_CHECK_ARG(loss_sum, "null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    DETBOX_HIP(hipMemsetAsync(loss_sum, 0, sizeof(double), s));
    if (rows == 0) return DETBOX_OK;
    ProfScope ps("smooth_l1_forward", s);
    smooth_l1_forward_kernel<<<dim3(loss_grid(rows)), dim3(LOSS_BLOCK), 0, s>>>(
        reinterpret_cast<const float4 *>(input), reinterpret_cast<const float4 *>(target), labels, (long long)rows, loss_sum);
    DETBOX_LAUNCH_CHECK();
    return DETBOX_OK;
}

extern "C" int detbox_smooth_l1_masked_backward(const float *input, const float *target, const int64_t *labels,
                                                int64_t rows, const float *grad_scale, float *grad_input, void *stream) {
    if (int rc = check_rows(input, target, labels, rows)) return rc;
    if (rows == 0) return DETBOX_OK;
    DETBOX_CHECK_ARG(grad_scale && grad_input, "null pointer");
    DETBOX_CHECK_ARG((uintptr_t)grad_input % 16 == 0, "box rows must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope ps("smooth_l1_backward", s);
    smooth_l1_backward_kernel<<<dim3(loss_grid(rows)), dim3(LOSS_BLOCK), 0, s>>>(
        reinterpret_cast<const float4 *>(input), reinterpret_cast<const float4 *>(target), labels, (long long)rows, grad_scale,
        reinterpret_cast<float4 *>(grad_input));
    DETBOX_LAUNCH_CHECK();
    return DETBOX_OK;
}
