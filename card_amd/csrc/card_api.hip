// card_api.hip - the C ABI of libcard_hip.so (include/card_hip.h): argument checking, dispatch to the two
// back ends (card_guided.hip: persistent MLP kernel; card_wide.hip: streaming-GEMM classification head),
// plus the small stand-alone entry points (coefficient table, q_sample, RNG fill).
#include <new>
#include <pthread.h>

#include "card_internal.h"

using namespace card;

namespace {
thread_local char g_err[512] = "";
__global__ void q_sample_kernel(const float* __restrict__ y, const float* __restrict__ y0hat,
                                const float* __restrict__ sab, const float* __restrict__ somab,
                                const int64_t* __restrict__ t, long long n_elems, int y_dim, int T,
                                const float* __restrict__ noise, unsigned long long seed,
                                unsigned long long elem_offset, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems;
         i += (long long)gridDim.x * blockDim.x) {
        const int64_t ti = t[i / y_dim];
        if (ti < 0 || ti >= T) {                 // torch.gather raises here; a kernel cannot: poison, never read out of bounds
            out[i] = __builtin_nanf("");
            continue;
        }
        const float a = sab[ti], s = somab[ti];
        float e;
        if (noise) {
            e = noise[i];
        } else {
            float z[4];
            philox_normal4(seed, elem_offset + (unsigned long long)i, 0u, z);
            e = z[0];
        }
        // regression/diffusion_utils.py:52, left to right, no contraction
        out[i] = (a * y[i] + (1.0f - a) * y0hat[i]) + s * e;
    }
}

__global__ void philox_fill_kernel(unsigned long long seed, unsigned long long elem_offset, long long n_elems,
                                   int draw0, int n_draws, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems;
         i += (long long)gridDim.x * blockDim.x) {
        float z[4];
        uint32_t blk = 0xFFFFFFFFu;
        for (int k = 0; k < n_draws; ++k) {
            const int d = draw0 + k;
            if (((uint32_t)d >> 2) != blk) {
                blk = (uint32_t)d >> 2;
                philox_normal4(seed, elem_offset + (unsigned long long)i, blk, z);
            }
            out[(long long)k * n_elems + i] = pick4(z, d & 3);
        }
    }
}

// Position-weighted 64-bit sum of the raw bits of up to kHashMax device arrays: order-independent across threads
// (so atomics are fine), sensitive to which element changed (a permutation or a swap of two tensors changes it).
constexpr int kHashMax = 32;
struct HashArgs {
    const uint32_t* ptr[kHashMax];
    long long count[kHashMax];
    int n;
};
__global__ __launch_bounds__(256) void content_hash_kernel(const HashArgs a, unsigned long long* __restrict__ out) {
    unsigned long long h = 0;
    for (int k = 0; k < a.n; ++k) {
        const uint32_t* p = a.ptr[k];
        const unsigned long long salt = 0x9E3779B97F4A7C15ull * (unsigned long long)(k + 1);
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.count[k]; i += (long long)gridDim.x * blockDim.x)
            h += ((unsigned long long)p[i] + 0x632BE59BD9B4E019ull) * ((unsigned long long)(2 * i + 1) * 0xD6E8FEB86659FD93ull + salt);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
    if ((threadIdx.x & 63) == 0 && h != 0) atomicAdd(out, h);
}

// ------------------------------------------------------------------------------------------------
// card_sampler_run_static: the library's own scratch, one copy per GPU, for callers that cannot allocate.
// `scratch` is the host-side NAME of a device array; its device address is asked for once per device
// (hipGetSymbolAddress with that device current) and only that pointer is ever handed to kernels.
// ------------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[CARD_STATIC_SCRATCH_BYTES];
constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices];
bool g_scratch_known[kMaxDevices];
pthread_mutex_t g_scratch_mu = PTHREAD_MUTEX_INITIALIZER;

}  // namespace

// The device address of `scratch` on the CURRENT device (looked up on the first call per device, then cached).
static int static_scratch(int device, void** out) {
    if (device < 0 || device >= kMaxDevices) return card_fail(CARD_E_UNSUPPORTED, "card_sampler_run_static: device index %d >= %d", device, kMaxDevices);
    pthread_mutex_lock(&g_scratch_mu);
    if (!g_scratch_known[device]) {
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            pthread_mutex_unlock(&g_scratch_mu);
            return card_fail(CARD_E_HIP, "card_sampler_run_static: hipGetSymbolAddress(scratch) on device %d: %s", device,
                             e != hipSuccess ? hipGetErrorString(e) : "null address");
        }
        g_scratch_ptr[device] = ptr;
        g_scratch_known[device] = true;
    }
    *out = g_scratch_ptr[device];
    pthread_mutex_unlock(&g_scratch_mu);
    return CARD_OK;
}

// One environment variable per back end, comma-separated tokens (card_internal.h: CardTune); the digits 0 / 2 of earlier
// rounds still mean off / force.
static bool env_has(const char* name, const char* token, char digit = 0) {
    const char* v = getenv(name);
    if (!v) return false;
    if (digit && v[0] == digit && (v[1] == 0 || v[1] == ',')) return true;
    const size_t n = strlen(token);
    for (const char* q = v; (q = strstr(q, token)) != nullptr; q += n)
        if ((q == v || q[-1] == ',') && (q[n] == 0 || q[n] == ',')) return true;
    return false;
}

void card_tune_from_env(CardTune* t) {
    memset(t, 0, sizeof(*t));
    t->coop_off = env_has("CARD_AMD_COOP", "off", '0');
    t->coop_force = env_has("CARD_AMD_COOP", "force", '2');
    t->coop_no16 = env_has("CARD_AMD_COOP", "no16");
    t->coop_no2 = env_has("CARD_AMD_COOP", "no2");
    t->stream_off = env_has("CARD_AMD_STREAM", "off", '0');
    t->stream_big = env_has("CARD_AMD_STREAM", "big");
    t->stream_nosplit = env_has("CARD_AMD_STREAM", "nosplit");
    t->stream_noslice = env_has("CARD_AMD_STREAM", "noslice");
    t->cls_off = env_has("CARD_AMD_CLSCOOP", "off", '0');
    t->cls_noslice = env_has("CARD_AMD_CLSCOOP", "noslice");
    t->wide_gemmhead = env_has("CARD_AMD_WIDE", "gemmhead");
    t->wide_notail = env_has("CARD_AMD_WIDE", "notail");
    t->arith_split = env_has("CARD_AMD_ARITH", "split");
}

int card_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define fail card_fail

extern "C" CARD_API int card_version(void) { return CARD_HIP_VERSION; }
extern "C" CARD_API const char* card_last_error(void) { return g_err; }

extern "C" CARD_API int card_make_coef_table(const float* alphas, const float* omabs, int32_t T, float* out) {
    if (!alphas || !omabs || !out || T < 1) return fail(CARD_E_INVALID, "card_make_coef_table: bad argument");
    // regression/diffusion_utils.py:71-80, :88 in the same fp32 operation order (file built with
    // -ffp-contract=off; sqrtf and / are correctly rounded, as torch's CPU kernels are)
    for (int t = 0; t < T; ++t) {
        float* r = out + (size_t)t * CARD_COEF_COLS;
        const float s = omabs[t];
        const float sab = sqrtf(1.0f - s * s);
        r[CARD_COEF_INV_SQRT_AB] = 1.0f / sab;
        r[CARD_COEF_ONE_M_SQRT_AB] = 1.0f - sab;
        r[CARD_COEF_SOMAB] = s;
        r[CARD_COEF_SQRT_AB] = sab;
        r[CARD_COEF_G0] = r[CARD_COEF_G1] = r[CARD_COEF_G2] = r[CARD_COEF_SIGMA] = 0.0f;
        if (t >= 1) {
            const float a = alphas[t];
            const float sm = omabs[t - 1];
            const float sabm = sqrtf(1.0f - sm * sm);
            const float s2 = s * s;
            r[CARD_COEF_G0] = (1.0f - a) * sabm / s2;
            r[CARD_COEF_G1] = (sm * sm) * sqrtf(a) / s2;
            r[CARD_COEF_G2] = 1.0f + (sab - 1.0f) * (sqrtf(a) + sabm) / s2;
            r[CARD_COEF_SIGMA] = sqrtf((sm * sm) / s2 * (1.0f - a));
        }
    }
    return CARD_OK;
}

extern "C" CARD_API int card_sampler_create(const card_model_desc* d, void* stream_, card_sampler** out) {
    if (!d || !out) return fail(CARD_E_INVALID, "card_sampler_create: null argument");
    *out = nullptr;
    if (d->struct_bytes != (int32_t)sizeof(card_model_desc))
        return fail(CARD_E_INVALID, "card_sampler_create: struct_bytes %d != %zu", d->struct_bytes, sizeof(card_model_desc));
    if (d->variant < CARD_VARIANT_GUIDED || d->variant > CARD_VARIANT_LINEAR_HADAMARD)
        return fail(CARD_E_INVALID, "card_sampler_create: unknown variant %d", d->variant);
    if (d->n_embed < 1 || d->x_dim < 0 || d->y_dim < 1 || d->hidden < 1) return fail(CARD_E_INVALID, "card_sampler_create: bad n_embed/x_dim/y_dim/hidden");
    for (int l = 0; l < (d->variant == CARD_VARIANT_LINEAR_HADAMARD ? 2 : 3); ++l)
        if (!d->W[l] || !d->b[l] || !d->gamma[l]) return fail(CARD_E_INVALID, "card_sampler_create: null weight pointer (layer %d)", l + 1);
    if (!d->W4 || !d->b4) return fail(CARD_E_INVALID, "card_sampler_create: null lin4 pointer");
    card_sampler* s = new (std::nothrow) card_sampler();
    if (!s) return fail(CARD_E_NOMEM, "card_sampler_create: out of host memory");
    memset(s, 0, sizeof(*s));
    s->variant = d->variant; s->x_dim = d->x_dim; s->y_dim = d->y_dim; s->hidden = d->hidden;
    s->n_embed = d->n_embed; s->cat_x = d->cat_x ? 1 : 0; s->cat_y_pred = d->cat_y_pred ? 1 : 0;
    s->d_in = d->variant == CARD_VARIANT_GUIDED
                  ? d->y_dim + (d->cat_y_pred ? d->y_dim : 0) + (d->cat_x ? d->x_dim : 0)
                  : d->y_dim * (d->cat_y_pred ? 2 : 1);   /* cat(y_t, y_0_hat) when `guidance` */
    hipError_t e;
    if ((e = hipGetDevice(&s->device)) != hipSuccess ||
        (e = hipDeviceGetAttribute(&s->num_cu, hipDeviceAttributeMultiprocessorCount, s->device)) != hipSuccess) {
        delete s;
        return fail(CARD_E_HIP, "card_sampler_create: %s", hipGetErrorString(e));
    }
    // the persistent kernel is built for the reference's regression network (H = 128, y_dim = 1); every other
    // shape (classification heads, wider or multi-output regression nets) streams through the GEMM back end
    card_tune_from_env(&s->tune);      // the only place the library reads the environment
    s->backend = (d->variant == CARD_VARIANT_GUIDED && d->hidden == 128 && d->y_dim == 1) ? 0
                 : (stream_supports(d, s->d_in, s->tune) ? 2 : (clscoop_supports(d, s->tune) ? 3 : 1));
    const int rc = s->backend == 0 ? guided_create(s, d, (hipStream_t)stream_)
                 : s->backend == 2 ? stream_create(s, d, (hipStream_t)stream_)
                 : s->backend == 3 ? clscoop_create(s, d, (hipStream_t)stream_) : wide_create(s, d, (hipStream_t)stream_);
    if (rc != CARD_OK) {
        card_sampler_destroy(s);
        return rc;
    }
    *out = s;
    return CARD_OK;
}

extern "C" CARD_API void card_sampler_destroy(card_sampler* s) {
    if (!s) return;
    if (s->backend == 0) guided_destroy(s);
    else if (s->backend == 2) stream_destroy(s);
    else if (s->backend == 3) clscoop_destroy(s);
    else wide_destroy(s);
    delete s;
}

extern "C" CARD_API int card_sampler_workspace_bytes(const card_sampler* s, int64_t N, uint64_t* bytes) {
    if (!s || !bytes || N < 0) return fail(CARD_E_INVALID, "card_sampler_workspace_bytes: bad argument");
    *bytes = s->backend == 1 ? wide_workspace_bytes(s, N) : 0;
    return CARD_OK;
}

// shared body of card_sampler_run / card_sampler_run_static: `lib_scratch` != NULL replaces every scratch buffer a
// back end would otherwise use (the sampler's own per-wave slots, the caller's workspace)
static int sampler_run(const card_sampler* s, const card_run_args* a, void* stream, void* lib_scratch, unsigned long long lib_scratch_bytes,
                       CardPlanRec* plan = nullptr) {
    if (!s || !a) return fail(CARD_E_INVALID, "card_sampler_run: null argument");
    if (a->struct_bytes != (int32_t)sizeof(card_run_args))
        return fail(CARD_E_INVALID, "card_sampler_run: struct_bytes %d != %zu", a->struct_bytes, sizeof(card_run_args));
    if (a->N < 0) return fail(CARD_E_INVALID, "card_sampler_run: N < 0");
    if (a->T < 1 || a->T > s->n_embed)
        return fail(CARD_E_INVALID, "card_sampler_run: T=%d outside [1, n_embed=%d]", a->T, s->n_embed);
    if (a->t_lo < 0 || a->t_hi < a->t_lo || a->t_hi >= a->T)
        return fail(CARD_E_INVALID, "card_sampler_run: need 0 <= t_lo <= t_hi < T (got %d, %d, T=%d)", a->t_lo, a->t_hi, a->T);
    if (!a->y_init && a->t_hi != a->T - 1)
        return fail(CARD_E_INVALID, "card_sampler_run: y_init == NULL starts a loop and needs t_hi == T-1");
    if (a->out_mode < CARD_OUT_LAST || a->out_mode > CARD_OUT_SELECT) return fail(CARD_E_INVALID, "card_sampler_run: bad out_mode");
    if (a->N == 0) return CARD_OK;  // empty batch: nothing to read or write
    if (a->n_z < 0 || (a->n_z > 1 && a->N % a->n_z != 0))
        return fail(CARD_E_INVALID, "card_sampler_run: N=%lld is not a multiple of n_z=%d", (long long)a->N, a->n_z);
    const float* y0hat = a->y_0_hat;
    const float* ytmean = a->y_T_mean;
    if (a->fphi) {
        // regression/card_regression.py:1021: y_0_hat = f_phi(x) over the un-tiled rows, first kernel of this call
        if (s->variant != CARD_VARIANT_GUIDED) return fail(CARD_E_UNSUPPORTED, "card_sampler_run: fphi is for CARD_VARIANT_GUIDED (regression) samplers");
        if (!a->x || !a->fphi_out) return fail(CARD_E_INVALID, "card_sampler_run: fphi needs x and fphi_out");
        if (fphi_out_dim(a->fphi) != s->y_dim)
            return fail(CARD_E_INVALID, "card_sampler_run: f_phi produces %d outputs, the sampler's y_dim is %d", fphi_out_dim(a->fphi), s->y_dim);
        if (s->cat_x && fphi_in_dim(a->fphi) != s->x_dim)
            return fail(CARD_E_INVALID, "card_sampler_run: f_phi reads %d features, eps_theta's x has %d", fphi_in_dim(a->fphi), s->x_dim);
        if (!card_plan_note(plan, "fphi_forward_kernel", false))
            if (int rc = fphi_forward(a->fphi, a->x, a->N / (a->n_z > 1 ? a->n_z : 1), a->fphi_out, (hipStream_t)stream)) return rc;
        y0hat = a->fphi_out;
        if (!ytmean) ytmean = a->fphi_out;
    }
    if (!y0hat || !ytmean || !a->coef || !a->out)
        return fail(CARD_E_INVALID, "card_sampler_run: null y_0_hat / y_T_mean / coef / out");
    if (((uintptr_t)a->coef & 15) != 0) return fail(CARD_E_INVALID, "card_sampler_run: coef must be 16-byte aligned (rows are read as two 16-byte loads)");
    const bool need_x = s->variant != CARD_VARIANT_GUIDED || (s->cat_x && s->x_dim > 0);
    if (need_x && !a->x) return fail(CARD_E_INVALID, "card_sampler_run: x is NULL but the model reads x");
    if (a->out_mode == CARD_OUT_SELECT && !a->state_slot) return fail(CARD_E_INVALID, "card_sampler_run: CARD_OUT_SELECT needs state_slot");
    CardRun r;
    memset(&r, 0, sizeof(r));
    r.x = a->x; r.y0hat = y0hat; r.ytmean = ytmean; r.y_init = a->y_init; r.coef = a->coef;
    r.noise = a->noise; r.out = a->out; r.state_slot = a->state_slot;
    r.workspace = a->workspace; r.workspace_bytes = a->workspace_bytes;
    r.N = a->N; r.seed = a->seed; r.elem_offset = a->elem_offset;
    r.T = a->T; r.t_hi = a->t_hi; r.t_lo = a->t_lo; r.out_mode = a->out_mode; r.eps_only = 0;
    r.n_z = a->n_z > 1 ? a->n_z : 1;
    if (a->arith != CARD_ARITH_F32_MFMA && a->arith != CARD_ARITH_SPLIT_BF16) return fail(CARD_E_INVALID, "card_sampler_run: unknown arith %d", a->arith);
    if (a->arith == CARD_ARITH_SPLIT_BF16 && s->backend != 0)
        return fail(CARD_E_UNSUPPORTED, "card_sampler_run: CARD_ARITH_SPLIT_BF16 is built for GUIDED samplers at hidden 128, y_dim 1");
    r.arith = a->arith;
    if (lib_scratch) {
        r.workspace = lib_scratch; r.workspace_bytes = lib_scratch_bytes; r.lib_scratch = 1;
    }
    r.plan = plan;
    return s->backend == 0 ? guided_run(s, r, (hipStream_t)stream)
         : s->backend == 2 ? stream_run(s, r, (hipStream_t)stream)
         : s->backend == 3 ? clscoop_run(s, r, (hipStream_t)stream) : wide_run(s, r, (hipStream_t)stream);
}

extern "C" CARD_API int card_sampler_run(const card_sampler* s, const card_run_args* a, void* stream) {
    return sampler_run(s, a, stream, nullptr, 0);
}

extern "C" CARD_API int card_sampler_plan(const card_sampler* s, const card_run_args* a, card_plan* out) {
    if (!s || !a || !out) return fail(CARD_E_INVALID, "card_sampler_plan: null argument");
    if (out->struct_bytes != (int32_t)sizeof(card_plan))
        return fail(CARD_E_INVALID, "card_sampler_plan: struct_bytes %d != %zu", out->struct_bytes, sizeof(card_plan));
    CardPlanRec rec;
    memset(&rec, 0, sizeof(rec));
    if (int rc = sampler_run(s, a, nullptr, nullptr, 0, &rec)) return rc;
    memset(out, 0, sizeof(*out));
    out->struct_bytes = (int32_t)sizeof(card_plan);
    out->backend = s->backend;
    out->launches = rec.launches;
    out->dominant_launches = rec.dominant_launches;
    memcpy(out->kernel, rec.kernel, sizeof(out->kernel));
    const double H = s->hidden;
    out->flops_per_sample_step = s->variant == CARD_VARIANT_LINEAR_HADAMARD ? 2.0 * H * (s->d_in + H + s->y_dim)
                                                                           : 2.0 * H * (s->d_in + 2.0 * H + s->y_dim);
    out->flops_executed_per_sample_step = rec.executed;
    return CARD_OK;
}

extern "C" CARD_API int card_sampler_run_static(const card_sampler* s, const card_run_args* a, void* stream) {
    if (!s || !a) return fail(CARD_E_INVALID, "card_sampler_run_static: null argument");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != s->device)
        return fail(CARD_E_INVALID, "card_sampler_run_static: the sampler lives on device %d but device %d is current", s->device, dev);
    void* ptr = nullptr;
    if (int rc = static_scratch(dev, &ptr)) return rc;
    return sampler_run(s, a, stream, ptr, CARD_STATIC_SCRATCH_BYTES);
}

extern "C" CARD_API int card_content_hash(const void* const* arrays, const int64_t* n_words, int32_t n_arrays, uint64_t* out,
                                          void* stream) {
    if (!arrays || !n_words || !out || n_arrays < 0 || n_arrays > kHashMax)
        return fail(CARD_E_INVALID, "card_content_hash: bad argument (at most %d arrays)", kHashMax);
    HashArgs h;
    memset(&h, 0, sizeof(h));
    long long total = 0;
    for (int k = 0; k < n_arrays; ++k) {
        if (n_words[k] < 0 || (n_words[k] > 0 && !arrays[k])) return fail(CARD_E_INVALID, "card_content_hash: array %d is null or negative-sized", k);
        h.ptr[k] = (const uint32_t*)arrays[k];
        h.count[k] = n_words[k];
        total += n_words[k];
    }
    h.n = n_arrays;
    HIP_TRY(hipMemsetAsync(out, 0, sizeof(uint64_t), (hipStream_t)stream));
    if (total == 0) return CARD_OK;
    long long grid = (total + 256 * 8 - 1) / (256 * 8);
    if (grid > 2048) grid = 2048;
    content_hash_kernel<<<dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream>>>(h, (unsigned long long*)out);
    HIP_TRY(hipGetLastError());
    return CARD_OK;
}

extern "C" CARD_API int card_eps_theta(const card_sampler* s, const float* x, const float* y_t, const float* y_0_hat,
                                       int64_t N, int32_t t, float* out, void* workspace, uint64_t workspace_bytes,
                                       void* stream) {
    if (!s || !y_t || !y_0_hat || !out) return fail(CARD_E_INVALID, "card_eps_theta: null argument");
    const bool need_x = s->variant != CARD_VARIANT_GUIDED || (s->cat_x && s->x_dim > 0);
    if (need_x && !x) return fail(CARD_E_INVALID, "card_eps_theta: x is NULL but the model reads x");
    if (t < 0 || t >= s->n_embed) return fail(CARD_E_INVALID, "card_eps_theta: t=%d outside the gamma table", t);
    if (N < 0) return fail(CARD_E_INVALID, "card_eps_theta: N < 0");
    if (N == 0) return CARD_OK;
    CardRun r;
    memset(&r, 0, sizeof(r));
    r.x = x; r.y0hat = y_0_hat; r.ytmean = y_0_hat; r.y_init = y_t; r.out = out; r.N = N;
    r.workspace = workspace; r.workspace_bytes = workspace_bytes;
    r.T = t + 1; r.t_hi = t; r.t_lo = t; r.out_mode = CARD_OUT_LAST; r.eps_only = 1; r.n_z = 1;
    return s->backend == 0 ? guided_run(s, r, (hipStream_t)stream)
         : s->backend == 2 ? stream_run(s, r, (hipStream_t)stream)
         : s->backend == 3 ? clscoop_run(s, r, (hipStream_t)stream) : wide_run(s, r, (hipStream_t)stream);
}

extern "C" CARD_API int card_q_sample(const float* y, const float* y_0_hat, const float* sab, const float* somab,
                             const int64_t* t, int64_t N, int32_t y_dim, int32_t T, const float* noise, uint64_t seed,
                             uint64_t elem_offset, float* out, void* stream) {
    if (!y || !y_0_hat || !sab || !somab || !t || !out) return fail(CARD_E_INVALID, "card_q_sample: null argument");
    if (N < 0 || y_dim < 1 || T < 1) return fail(CARD_E_INVALID, "card_q_sample: bad N / y_dim / T");
    if (N == 0) return CARD_OK;
    const long long n = (long long)N * y_dim;
    long long grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    q_sample_kernel<<<dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream>>>(y, y_0_hat, sab, somab, t, n, y_dim, T, noise,
                                                                                  seed, elem_offset, out);
    HIP_TRY(hipGetLastError());
    return CARD_OK;
}

extern "C" CARD_API int card_philox_normal(uint64_t seed, uint64_t elem_offset, int64_t n_elems, int32_t draw0, int32_t n_draws,
                                  float* out, void* stream) {
    if (!out || n_elems < 0 || draw0 < 0 || n_draws < 0) return fail(CARD_E_INVALID, "card_philox_normal: bad argument");
    if (n_elems == 0 || n_draws == 0) return CARD_OK;
    long long grid = (n_elems + 255) / 256;
    if (grid > 4096) grid = 4096;
    philox_fill_kernel<<<dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream>>>(seed, elem_offset, n_elems, draw0, n_draws, out);
    HIP_TRY(hipGetLastError());
    return CARD_OK;
}
