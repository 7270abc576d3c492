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
