// card_fphi.hip - the guidance network f_phi inside the sampling call.
//
// Reference: regression/model.py:55-82 DeterministicFeedForwardNeuralNetwork (Linear -> [BatchNorm1d] -> LeakyReLU ->
// Dropout per hidden layer, then a plain Linear; 13 -> 100 -> 50 -> 1 in the UCI configs) and the OLS guidance
// `cat(1, x).mm(beta)` (regression/card_regression.py:111-124), both evaluated once per test minibatch at
// regression/card_regression.py:1021 before the rows are tiled n_z times (:1027-1035).
//
// Here the network runs as the first kernel of card_sampler_run when card_run_args.fphi is set: one pass over the
// UN-tiled test rows writes y_0_hat (rows, y_dim), and the sampling kernels expand every row to its n_z chains
// themselves (card_run_args.n_z), so x is read once per row and the tiled copies of x / y_0_hat / y_T_mean of the
// reference are never materialised.  The work is ~6 k MACs per row against ~35 M per chain-loop: not a hot spot,
// so the kernel is a plain LDS-staged MLP (eval mode: BatchNorm folded to scale/shift, Dropout = identity).
#include <new>

#include "card_internal.h"

struct card_fphi {
    int n_layers;
    int dims[CARD_FPHI_MAX_LAYERS + 1];
    int max_dim;
    int device;
    float slope;
    float* Wt[CARD_FPHI_MAX_LAYERS];      // (in, out): transposed copies, so that consecutive threads read consecutive words
    float* b[CARD_FPHI_MAX_LAYERS];       // (out) or NULL
    float* scale[CARD_FPHI_MAX_LAYERS];   // folded eval-BatchNorm or NULL
    float* shift[CARD_FPHI_MAX_LAYERS];
};

namespace {

constexpr int kRows = 8;         // test rows per workgroup pass
constexpr int kThreadsF = 256;

struct FphiParams {
    int n_layers;
    int dims[CARD_FPHI_MAX_LAYERS + 1];
    int max_dim;
    float slope;
    const float* Wt[CARD_FPHI_MAX_LAYERS];
    const float* b[CARD_FPHI_MAX_LAYERS];
    const float* scale[CARD_FPHI_MAX_LAYERS];
    const float* shift[CARD_FPHI_MAX_LAYERS];
};

__global__ void transpose_kernel(const float* __restrict__ W, int out_f, int in_f, float* __restrict__ Wt) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= out_f * in_f) return;
    const int o = idx / in_f, k = idx % in_f;
    Wt[(long long)k * out_f + o] = W[idx];
}

__global__ __launch_bounds__(kThreadsF) void fphi_forward_kernel(const FphiParams p, const float* __restrict__ x,
                                                                 long long rows, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float act[];      // two [kRows][max_dim] activation buffers
    const int tid = threadIdx.x;
    const long long n_blocks = (rows + kRows - 1) / kRows;
    for (long long blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const long long row0 = blk * kRows;
        const int nr = rows - row0 < kRows ? (int)(rows - row0) : kRows;
        float* cur = act;
        float* nxt = act + kRows * p.max_dim;
        const int d0 = p.dims[0];
        for (int i = tid; i < nr * d0; i += kThreadsF) cur[i] = x[row0 * d0 + i];     // rows are contiguous: coalesced
        __syncthreads();
        for (int l = 0; l < p.n_layers; ++l) {
            const int in_f = p.dims[l], out_f = p.dims[l + 1];
            const bool last = l == p.n_layers - 1;
            const float* __restrict__ Wt = p.Wt[l];
            for (int idx = tid; idx < nr * out_f; idx += kThreadsF) {
                const int r = idx / out_f, o = idx % out_f;
                const float* a = cur + r * in_f;
                float acc = 0.0f;
                for (int k = 0; k < in_f; ++k) acc = fmaf(a[k], Wt[(long long)k * out_f + o], acc);
                if (p.b[l]) acc += p.b[l][o];
                if (!last) {
                    if (p.scale[l]) acc = acc * p.scale[l][o] + p.shift[l][o];         // eval-mode BatchNorm1d, model.py:74-75
                    acc = acc > 0.0f ? acc : acc * p.slope;                            // LeakyReLU, model.py:76
                    nxt[r * out_f + o] = acc;
                } else {
                    out[(row0 + r) * out_f + o] = acc;
                }
            }
            __syncthreads();
            float* t = cur; cur = nxt; nxt = t;
        }
    }
}

}  // namespace

extern "C" CARD_API int card_fphi_create(const card_fphi_desc* d, void* stream_, card_fphi** out) {
    if (!d || !out) return card_fail(CARD_E_INVALID, "card_fphi_create: null argument");
    *out = nullptr;
    if (d->struct_bytes != (int32_t)sizeof(card_fphi_desc))
        return card_fail(CARD_E_INVALID, "card_fphi_create: struct_bytes %d != %zu", d->struct_bytes, sizeof(card_fphi_desc));
    if (d->n_layers < 1 || d->n_layers > CARD_FPHI_MAX_LAYERS)
        return card_fail(CARD_E_UNSUPPORTED, "card_fphi_create: n_layers=%d outside [1, %d]", d->n_layers, CARD_FPHI_MAX_LAYERS);
    int max_dim = 0;
    for (int l = 0; l <= d->n_layers; ++l) {
        if (d->dims[l] < 1 || d->dims[l] > CARD_FPHI_MAX_WIDTH)
            return card_fail(CARD_E_UNSUPPORTED, "card_fphi_create: dims[%d]=%d outside [1, %d]", l, d->dims[l], CARD_FPHI_MAX_WIDTH);
        if (d->dims[l] > max_dim) max_dim = d->dims[l];
    }
    for (int l = 0; l < d->n_layers; ++l) {
        if (!d->W[l]) return card_fail(CARD_E_INVALID, "card_fphi_create: null weight pointer (layer %d)", l);
        if ((d->bn_scale[l] == nullptr) != (d->bn_shift[l] == nullptr))
            return card_fail(CARD_E_INVALID, "card_fphi_create: layer %d has only one of bn_scale / bn_shift", l);
    }
    card_fphi* f = new (std::nothrow) card_fphi();
    if (!f) return card_fail(CARD_E_NOMEM, "card_fphi_create: out of host memory");
    memset(f, 0, sizeof(*f));
    f->n_layers = d->n_layers; f->max_dim = max_dim; f->slope = d->negative_slope;
    for (int l = 0; l <= d->n_layers; ++l) f->dims[l] = d->dims[l];
    hipStream_t stream = (hipStream_t)stream_;
    auto fail_free = [&](hipError_t e, const char* what) {
        card_fphi_destroy(f);
        return card_fail(CARD_E_HIP, "card_fphi_create: %s: %s", what, hipGetErrorString(e));
    };
    hipError_t e = hipGetDevice(&f->device);
    if (e != hipSuccess) return fail_free(e, "hipGetDevice");
    for (int l = 0; l < d->n_layers; ++l) {
        const int in_f = d->dims[l], out_f = d->dims[l + 1];
        const size_t wb = (size_t)in_f * out_f * sizeof(float), vb = (size_t)out_f * sizeof(float);
        if ((e = hipMalloc(&f->Wt[l], wb)) != hipSuccess) return fail_free(e, "hipMalloc");
        transpose_kernel<<<(in_f * out_f + 255) / 256, 256, 0, stream>>>(d->W[l], out_f, in_f, f->Wt[l]);
        const float* src[3] = {d->b[l], d->bn_scale[l], d->bn_shift[l]};
        float** dst[3] = {&f->b[l], &f->scale[l], &f->shift[l]};
        for (int i = 0; i < 3; ++i) {
            if (!src[i]) continue;
            if ((e = hipMalloc(dst[i], vb)) != hipSuccess) return fail_free(e, "hipMalloc");
            if ((e = hipMemcpyAsync(*dst[i], src[i], vb, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return fail_free(e, "hipMemcpyAsync");
        }
    }
    if ((e = hipGetLastError()) != hipSuccess) return fail_free(e, "transpose launch");
    *out = f;
    return CARD_OK;
}

extern "C" CARD_API void card_fphi_destroy(card_fphi* f) {
    if (!f) return;
    for (int l = 0; l < CARD_FPHI_MAX_LAYERS; ++l) {
        if (f->Wt[l]) (void)hipFree(f->Wt[l]);
        if (f->b[l]) (void)hipFree(f->b[l]);
        if (f->scale[l]) (void)hipFree(f->scale[l]);
        if (f->shift[l]) (void)hipFree(f->shift[l]);
    }
    delete f;
}

int fphi_in_dim(const card_fphi* f) { return f->dims[0]; }
int fphi_out_dim(const card_fphi* f) { return f->dims[f->n_layers]; }

int fphi_forward(const card_fphi* f, const float* x, long long rows, float* out, hipStream_t stream) {
    FphiParams p;
    memset(&p, 0, sizeof(p));
    p.n_layers = f->n_layers; p.max_dim = f->max_dim; p.slope = f->slope;
    for (int l = 0; l <= f->n_layers; ++l) p.dims[l] = f->dims[l];
    for (int l = 0; l < f->n_layers; ++l) {
        p.Wt[l] = f->Wt[l]; p.b[l] = f->b[l]; p.scale[l] = f->scale[l]; p.shift[l] = f->shift[l];
    }
    long long grid = (rows + kRows - 1) / kRows;
    if (grid > 8192) grid = 8192;
    const size_t lds = (size_t)2 * kRows * f->max_dim * sizeof(float);      // <= 64 KB at CARD_FPHI_MAX_WIDTH
    fphi_forward_kernel<<<dim3((unsigned)grid), dim3(kThreadsF), lds, stream>>>(p, x, rows, out);
    HIP_TRY(hipGetLastError());
    return CARD_OK;
}

extern "C" CARD_API int card_fphi_forward(const card_fphi* f, const float* x, int64_t rows, float* out, void* stream) {
    if (!f || rows < 0) return card_fail(CARD_E_INVALID, "card_fphi_forward: bad argument");
    if (rows == 0) return CARD_OK;
    if (!x || !out) return card_fail(CARD_E_INVALID, "card_fphi_forward: null x / out");
    return fphi_forward(f, x, rows, out, (hipStream_t)stream);
}
