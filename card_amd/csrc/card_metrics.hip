// card_metrics.hip - per-row evaluation statistics of generated regression samples, on the device.
//
// Replaces the host/numpy side of the reference's test loop (regression/card_regression.py): for every test
// row it needs the n_z generated samples sorted (np.percentile with linear interpolation at 0,10,...,100 for
// QICE :786-811 and at PICP_range :813-825, the trimmed-mean prediction :767-784) plus a logsumexp NLL
// (:866-893).  The reference copies (T+1) x N samples to the host for this; here one workgroup per
// (state, row) sorts the row's samples in LDS (bitonic, n_z padded to a power of two) and writes four numbers,
// so only rows x 4 values per state leave the GPU (or are summed and all-reduced across GPUs).
#include "card_internal.h"

namespace {

constexpr int kMaxNz = 8192;   // LDS: 32 KB of samples per row

struct MetricParams {
    const float* samples;   // (n_states, rows, n_z)
    const float* y_true;    // (rows)  normalised targets
    long long rows;
    int n_z, n_pad, n_bins;
    int trim_lo_i, trim_hi_i; // sorted-sample index window of the trimmed mean; lo == hi: median (trim range (50, 50))
    float y_scale, y_shift; // un-normalise: v * scale + shift
    float picp_lo, picp_hi;
    float tau;              // > 0: global precision; <= 0: per-row 1 / var(samples, unbiased) of un-normalised samples
    float* se;              // (n_states, rows) squared error of the (trimmed) mean prediction, un-normalised
    float* nll;             // (n_states, rows)
    int* qice_bin;          // (n_states, rows) number of decile boundaries below y_true: 0 .. n_bins + 1
    int* picp_in;           // (n_states, rows) 1 if y_true inside [P_lo, P_hi] of the samples
};

// np.percentile(..., method="linear") on a sorted row
__device__ __forceinline__ double percentile_sorted(const float* s, int n, double q) {
    const double pos = q / 100.0 * (double)(n - 1);
    int lo = (int)floor(pos);
    if (lo < 0) lo = 0;
    if (lo > n - 1) lo = n - 1;
    const int hi = lo + 1 < n ? lo + 1 : n - 1;
    const double t = pos - (double)lo;
    const double a = s[lo], b = s[hi];
    // numpy's _lerp: a + (b - a) * t for t < 0.5, b - (b - a) * (1 - t) otherwise
    return t < 0.5 ? a + (b - a) * t : b - (b - a) * (1.0 - t);
}

__global__ __launch_bounds__(256) void row_metrics_kernel(const MetricParams p) {
    extern __shared__ float srt[];             // n_pad floats
    __shared__ double red[256];
    const long long row = blockIdx.x;
    const long long st = blockIdx.y;
    const float* src = p.samples + (st * p.rows + row) * p.n_z;
    const int tid = threadIdx.x, n = p.n_z, P = p.n_pad;
    for (int i = tid; i < P; i += 256) srt[i] = i < n ? src[i] : __builtin_inff();
    __syncthreads();
    // NLL pieces before sorting order matters nowhere; use the unsorted copy in LDS
    const float yt_n = p.y_true[row];
    const double yt = (double)yt_n * p.y_scale + p.y_shift;
    // mean / variance of un-normalised samples (for tau = 1/var when requested)
    double s1 = 0.0, s2 = 0.0;
    for (int i = tid; i < n; i += 256) {
        const double v = (double)srt[i] * p.y_scale + p.y_shift;
        s1 += v; s2 += v * v;
    }
    red[tid] = s1; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const double sum1 = red[0]; __syncthreads();
    red[tid] = s2; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const double sum2 = red[0]; __syncthreads();
    double tau = p.tau;
    if (!(tau > 0.0)) {
        const double mean = sum1 / n;
        const double var = n > 1 ? (sum2 - n * mean * mean) / (n - 1) : 1.0;
        tau = 1.0 / var;
    }
    // logsumexp_z(-0.5 tau (y - y_z)^2): max then sum
    double mx = -1e300;
    for (int i = tid; i < n; i += 256) {
        const double d = yt - ((double)srt[i] * p.y_scale + p.y_shift);
        const double e = -0.5 * tau * d * d;
        mx = e > mx ? e : mx;
    }
    red[tid] = mx; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = red[tid] > red[tid + o] ? red[tid] : red[tid + o]; __syncthreads(); }
    const double gmx = red[0]; __syncthreads();
    double se_sum = 0.0;
    for (int i = tid; i < n; i += 256) {
        const double d = yt - ((double)srt[i] * p.y_scale + p.y_shift);
        se_sum += exp(-0.5 * tau * d * d - gmx);
    }
    red[tid] = se_sum; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const double lse = gmx + log(red[0]); __syncthreads();

    // bitonic sort of the padded row
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const float a = srt[i], b = srt[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { srt[i] = b; srt[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    // prediction: trimmed mean of the sorted samples (card_regression.py:773-776), or their median when the trim range
    // is (50, 50) (:770-771; np.median = mean of the two middle samples for even n) -- the host passes lo == hi then
    const int lo_i = p.trim_lo_i, hi_i = p.trim_hi_i;
    double tm = 0.0;
    for (int i = lo_i + tid; i < hi_i; i += 256) tm += srt[i];
    red[tid] = tm; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) {
        const long long o = st * p.rows + row;
        const double center = hi_i > lo_i ? red[0] / (double)(hi_i - lo_i)
                                          : ((n & 1) ? (double)srt[n / 2] : 0.5 * ((double)srt[n / 2 - 1] + (double)srt[n / 2]));
        const double pred = center * p.y_scale + p.y_shift;
        p.se[o] = (float)((pred - yt) * (pred - yt));
        p.nll[o] = (float)(-(lse - log((double)n) - 0.5 * log(2.0 * 3.14159265358979323846) + 0.5 * log(tau)));
        int member = 0;
        for (int b = 0; b <= p.n_bins; ++b)
            member += ((double)yt_n - percentile_sorted(srt, n, 100.0 * b / p.n_bins)) > 0.0 ? 1 : 0;
        p.qice_bin[o] = member;
        const double ql = percentile_sorted(srt, n, p.picp_lo), qh = percentile_sorted(srt, n, p.picp_hi);
        p.picp_in[o] = ((double)yt_n >= ql && (double)yt_n <= qh) ? 1 : 0;
    }
}

}  // namespace

extern "C" CARD_API int card_regression_row_metrics(const card_metric_args* a, void* stream) {
    if (!a || a->struct_bytes != (int32_t)sizeof(card_metric_args))
        return card_fail(CARD_E_INVALID, "card_regression_row_metrics: bad struct_bytes");
    if (a->rows < 0 || a->n_states < 0 || a->n_z < 1 || a->n_bins < 1)
        return card_fail(CARD_E_INVALID, "card_regression_row_metrics: bad rows / n_states / n_z / n_bins");
    if (a->n_z > kMaxNz) return card_fail(CARD_E_UNSUPPORTED, "card_regression_row_metrics: n_z=%d > %d", a->n_z, kMaxNz);
    if (a->rows == 0 || a->n_states == 0) return CARD_OK;
    if (!a->samples || !a->y_true || !a->se || !a->nll || !a->qice_bin || !a->picp_in)
        return card_fail(CARD_E_INVALID, "card_regression_row_metrics: null pointer");
    if (a->n_states > 65535) return card_fail(CARD_E_UNSUPPORTED, "card_regression_row_metrics: n_states > 65535");
    MetricParams p;
    p.samples = a->samples; p.y_true = a->y_true; p.rows = a->rows; p.n_z = a->n_z; p.n_bins = a->n_bins;
    int P = 2;
    while (P < a->n_z) P <<= 1;
    p.n_pad = P;
    p.y_scale = a->y_scale; p.y_shift = a->y_shift;
    // index window of the trimmed mean exactly as the reference forms it: int(low / 100 * n_z) (card_regression.py:774-775)
    if (a->trim_lo == 50.0f && a->trim_hi == 50.0f) {
        p.trim_lo_i = p.trim_hi_i = 0;                 // median
    } else {
        p.trim_lo_i = (int)((double)a->trim_lo / 100.0 * a->n_z);
        p.trim_hi_i = (int)((double)a->trim_hi / 100.0 * a->n_z);
        if (!(a->trim_lo >= 0.0f && a->trim_hi <= 100.0f) || p.trim_hi_i <= p.trim_lo_i)
            return card_fail(CARD_E_INVALID, "card_regression_row_metrics: trimmed_mean_range (%g, %g) keeps no sample of n_z=%d "
                             "(the reference would average an empty slice); only (50, 50) selects the median",
                             (double)a->trim_lo, (double)a->trim_hi, a->n_z);
    }
    p.picp_lo = a->picp_lo; p.picp_hi = a->picp_hi; p.tau = a->tau;
    p.se = a->se; p.nll = a->nll; p.qice_bin = a->qice_bin; p.picp_in = a->picp_in;
    // gridDim.x is limited to 2^31-1 rows; rows beyond that would need chunking
    if (a->rows > 2147483647LL) return card_fail(CARD_E_UNSUPPORTED, "card_regression_row_metrics: too many rows");
    row_metrics_kernel<<<dim3((unsigned)a->rows, (unsigned)a->n_states), 256, (size_t)P * sizeof(float), (hipStream_t)stream>>>(p);
    HIP_TRY(hipGetLastError());
    return CARD_OK;
}

// ------------------------------------------------------------------------------------------------
// Classification: per-image statistics of the n_z generated y vectors (compute_and_store_cls_metrics,
// classification/card_classification.py:540-603), one workgroup per (state, image).
// LDS: A[C][n_pad] raw values -(y - 1)^2, class-major (sorted per class at the end for the credible interval),
// P[C][n_z] softmax probabilities.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int kClsMaxNz = 1024, kClsMaxC = 1024;
constexpr size_t kClsMaxLds = 144 * 1024;

struct ClsMetricParams {
    const float* samples;   // (n_states, rows, n_z, C)
    const int* labels;      // (rows)
    long long rows;
    int n_z, n_pad, C;
    float scale_T, q_lo, q_hi, alpha;     // q: fractions as float32, like torch.tensor([low / 100, high / 100])
    float* label_probs; float* mean_probs; float* ci; int* majority; float* inst_acc; double* pvalue; int* reject;
    float* sample_probs;
};

// continued fraction of the incomplete beta function (modified Lentz)
__device__ double betacf(double a, double b, double x) {
    const double tiny = 1e-300, eps = 1e-16;
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 500; ++m) {
        const double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < eps) break;
    }
    return h;
}

// regularised incomplete beta I_x(a, b)
__device__ double betainc(double a, double b, double x) {
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    const double bt = exp(lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log1p(-x));
    if (x < (a + 1.0) / (a + b + 2.0)) return bt * betacf(a, b, x) / a;
    return 1.0 - bt * betacf(b, a, 1.0 - x) / b;
}

// torch.quantile / nanquantile (linear) on a sorted float32 row: ranks and lerp in float32 (aten/native/Sorting.cpp)
__device__ __forceinline__ float quantile_sorted_f32(const float* s, int n, float q) {
    const float rank = q * (float)(n - 1);
    const float below = floorf(rank), w = rank - below;
    int lo = (int)below, hi = (int)ceilf(rank);
    lo = lo < 0 ? 0 : (lo > n - 1 ? n - 1 : lo);
    hi = hi < 0 ? 0 : (hi > n - 1 ? n - 1 : hi);
    const float a = s[lo], b = s[hi], diff = b - a;
    return w < 0.5f ? a + w * diff : b - diff * (1.0f - w);
}

__global__ __launch_bounds__(256) void cls_row_metrics_kernel(const ClsMetricParams p) {
    extern __shared__ float sm[];
    __shared__ int votes[kClsMaxC];
    __shared__ double red[256];
    __shared__ int top2[2];
    __shared__ double tmean;
    const int tid = threadIdx.x, n = p.n_z, P = p.n_pad, C = p.C;
    float* const A = sm;                    // [C][P]
    float* const Pr = sm + (size_t)C * P;   // [C][n]
    const long long row = blockIdx.x, o = (long long)blockIdx.y * p.rows + row;
    const float* src = p.samples + o * n * C;
    for (int idx = tid; idx < n * C; idx += 256) {
        const int i = idx / C, c = idx - i * C;
        const float d = src[idx] - 1.0f;
        A[c * P + i] = -(d * d);                                   // compute_val_before_softmax, :529-536
    }
    for (int idx = tid; idx < C * (P - n); idx += 256) {
        const int c = idx / (P - n), i = n + idx - c * (P - n);
        A[c * P + i] = __builtin_inff();
    }
    for (int c = tid; c < C; c += 256) votes[c] = 0;
    __syncthreads();
    // softmax(raw / scale_T) per sample (:553), its arg-max label (:562), votes
    for (int i = tid; i < n; i += 256) {
        float mx = -__builtin_inff();
        int lab = 0;
        for (int c = 0; c < C; ++c) {
            const float v = A[c * P + i] / p.scale_T;
            if (v > mx) { mx = v; lab = c; }
        }
        float sum = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float e = expf(A[c * P + i] / p.scale_T - mx);
            Pr[c * n + i] = e;
            sum += e;
        }
        for (int c = 0; c < C; ++c) Pr[c * n + i] = Pr[c * n + i] / sum;
        atomicAdd(&votes[lab], 1);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = 0.0f;
        for (int i = 0; i < n; ++i) s += Pr[c * n + i];
        p.label_probs[o * C + c] = (float)votes[c] / (float)n;      // :566-569
        p.mean_probs[o * C + c] = s / (float)n;                     // :570
    }
    if (p.sample_probs)
        for (int idx = tid; idx < n * C; idx += 256) {
            const int i = idx / C, c = idx - i * C;
            p.sample_probs[o * n * C + idx] = Pr[c * n + i];
        }
    if (tid == 0) {
        // majority vote (:572) and the runner-up of the t-test (:576-577); equal counts: lower class index first
        int m1 = 0;
        for (int c = 1; c < C; ++c) if (votes[c] > votes[m1]) m1 = c;
        int m2 = m1 == 0 ? (C > 1 ? 1 : 0) : 0;
        for (int c = 0; c < C; ++c) if (c != m1 && votes[c] > votes[m2]) m2 = c;
        top2[0] = m1; top2[1] = m2;
        p.majority[o] = m1;
        const int lbl = p.labels[row];
        p.inst_acc[o] = (lbl >= 0 && lbl < C) ? (float)votes[lbl] / (float)n : 0.0f;   // :574
    }
    __syncthreads();
    // paired two-sided t-test between the two most voted classes (:576-596): scipy.stats.ttest_rel on float32 inputs
    // forms d = (a - b) in float32 and continues in float64
    const int m1 = top2[0], m2 = top2[1];
    double s1 = 0.0;
    for (int i = tid; i < n; i += 256) s1 += (double)(Pr[m1 * n + i] - Pr[m2 * n + i]);
    red[tid] = s1; __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (tid < k) red[tid] += red[tid + k]; __syncthreads(); }
    if (tid == 0) tmean = red[0] / (double)n;
    __syncthreads();
    const double dm = tmean;
    double s2 = 0.0;
    for (int i = tid; i < n; i += 256) {
        const double d = (double)(Pr[m1 * n + i] - Pr[m2 * n + i]) - dm;
        s2 += d * d;
    }
    __syncthreads();
    red[tid] = s2; __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (tid < k) red[tid] += red[tid + k]; __syncthreads(); }
    if (tid == 0) {
        const double df = (double)(n - 1);
        const double var = n > 1 ? red[0] / df : __builtin_nan("");
        const double denom = sqrt(var / (double)n);
        const double t = dm / denom;                                 // 0 / 0 = NaN, x / 0 = +-inf, as numpy
        double pv;
        if (t != t) pv = __builtin_nan("");
        else if (isinf(t)) pv = 0.0;
        else pv = betainc(0.5 * df, 0.5, df / (df + t * t));         // = 2 sf(|t|, df)
        p.pvalue[o] = pv;
        p.reject[o] = pv < (double)p.alpha ? 1 : 0;                  // NaN compares false (:596)
    }
    // credible interval of the raw values per class (:556-560): bitonic sort of every class row at once
    const int total = C * P;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int idx = tid; idx < total; idx += 256) {
                const int i = idx & (P - 1), ixj = i ^ j;
                if (ixj > i) {
                    const int base = idx - i;
                    const float a = A[base + i], b = A[base + ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { A[base + i] = b; A[base + ixj] = a; }
                }
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        p.ci[(o * 2 + 0) * C + c] = quantile_sorted_f32(A + c * P, n, p.q_lo);
        p.ci[(o * 2 + 1) * C + c] = quantile_sorted_f32(A + c * P, n, p.q_hi);
    }
}

}  // namespace

extern "C" CARD_API int card_classification_row_metrics(const card_cls_metric_args* a, void* stream) {
    if (!a || a->struct_bytes != (int32_t)sizeof(card_cls_metric_args))
        return card_fail(CARD_E_INVALID, "card_classification_row_metrics: bad struct_bytes");
    if (a->rows < 0 || a->n_states < 0 || a->n_z < 1 || a->n_classes < 1)
        return card_fail(CARD_E_INVALID, "card_classification_row_metrics: bad rows / n_states / n_z / n_classes");
    if (a->n_z > kClsMaxNz || a->n_classes > kClsMaxC)
        return card_fail(CARD_E_UNSUPPORTED, "card_classification_row_metrics: n_z=%d / n_classes=%d beyond %d / %d", a->n_z, a->n_classes, kClsMaxNz, kClsMaxC);
    int P = 2;
    while (P < a->n_z) P <<= 1;
    const size_t lds = (size_t)a->n_classes * (size_t)(P + a->n_z) * sizeof(float);
    if (lds > kClsMaxLds)
        return card_fail(CARD_E_UNSUPPORTED, "card_classification_row_metrics: n_classes=%d x n_z=%d needs %zu bytes of LDS (limit %zu)",
                         a->n_classes, a->n_z, lds, kClsMaxLds);
    if (!(a->scale_T > 0.0f) || !(a->ci_lo >= 0.0f && a->ci_lo <= a->ci_hi && a->ci_hi <= 100.0f))
        return card_fail(CARD_E_INVALID, "card_classification_row_metrics: scale_T must be positive, 0 <= ci_lo <= ci_hi <= 100");
    if (a->rows == 0 || a->n_states == 0) return CARD_OK;
    if (!a->samples || !a->labels || !a->label_probs || !a->mean_probs || !a->ci || !a->majority_vote || !a->instance_accuracy ||
        !a->ttest_pvalue || !a->ttest_reject)
        return card_fail(CARD_E_INVALID, "card_classification_row_metrics: null pointer");
    if (a->n_states > 65535 || a->rows > 2147483647LL)
        return card_fail(CARD_E_UNSUPPORTED, "card_classification_row_metrics: too many states / rows for one launch");
    if (lds > 48 * 1024)    // per device and idempotent, a few microseconds
        HIP_TRY(hipFuncSetAttribute((const void*)cls_row_metrics_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kClsMaxLds));
    ClsMetricParams p;
    p.samples = a->samples; p.labels = a->labels; p.rows = a->rows; p.n_z = a->n_z; p.n_pad = P; p.C = a->n_classes;
    p.scale_T = a->scale_T; p.q_lo = (float)((double)a->ci_lo / 100.0); p.q_hi = (float)((double)a->ci_hi / 100.0); p.alpha = a->ttest_alpha;
    p.label_probs = a->label_probs; p.mean_probs = a->mean_probs; p.ci = a->ci; p.majority = a->majority_vote;
    p.inst_acc = a->instance_accuracy; p.pvalue = a->ttest_pvalue; p.reject = a->ttest_reject; p.sample_probs = a->sample_probs;
    cls_row_metrics_kernel<<<dim3((unsigned)a->rows, (unsigned)a->n_states), 256, lds, (hipStream_t)stream>>>(p);
    HIP_TRY(hipGetLastError());
    return CARD_OK;
}
