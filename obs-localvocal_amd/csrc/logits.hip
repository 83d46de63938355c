// logits.hip — K9/K10: fused logits rules + samplers on the device.
//
// Replaces whisper.cpp's whisper_process_logits + whisper_sample_token[_topk], which
// run on the host per decoder per step inside whisper_full
// (/root/reference/src/whisper-utils/whisper-processing.cpp:218-219).
//
// A row of n_vocab logits is cut into PG contiguous slabs, one 256-thread workgroup per (row, slab), so a
// beam step (5 rows) runs on 80 compute units instead of 5.  The reference's passes depend on row-wide
// values (maximum -> logsumexp -> timestamp mass -> probabilities), so the work is four short kernels that
// hand 64-byte slab records to each other; every workgroup of the next kernel folds the PG records of its
// row in slab order (deterministic):
//   A. temperature + suppression rules (static mask, blank, timestamp pairing, max_initial_ts, monotonic
//      timestamps), masked row written once to `work`, row / text / timestamp maxima per slab;
//   B. sum exp(v - max) per slab                                    (-> logsumexp);
//   C. per slab: timestamp logsumexp term, first-max argmax and probability mass (double) of the text and of
//      the timestamp tokens separately — the "timestamp mass > best text" rule only masks text tokens, so
//      both outcomes are covered by keeping the two classes apart;
//   D. decision of that rule, greedy result / timestamp statistics, optional full-row dumps, and (sampling
//      only) inverse-CDF draws from host-supplied uniforms — the distribution std::discrete_distribution
//      realises in the reference (normalised doubles, lower_bound): slab totals give each slab's base, the
//      slab whose interval holds the uniform scans its 3 242 entries.
// Only a 100-byte record per row crosses PCIe per step.  Bound: latency (n_vocab * 4 B per row from L2).
#include "kernels.hpp"

namespace mw {

namespace {

constexpr int PT = 256;      // threads per slab workgroup
constexpr int PW = PT / 64;  // waves
constexpr int PG = 16;       // slabs per row
constexpr int PV = 13;       // entries per thread: PG * PT * PV = 53 248 >= any Whisper vocabulary
constexpr int MAX_ROWS = 32;

struct SlabRec {                // one per (row, slab); written field group by field group by kernels A, B, C
    float mx, mx_text, mx_ts;   // A
    float se;                   // B
    float st;                   // C: sum over live timestamps of exp(logprob - max timestamp logprob)
    float best_p, best_lp;      // C: text tokens, first maximum
    int best_i;
    float best_tp, best_tlp;    // C: timestamp tokens, first maximum
    int best_ti;
    int pad;
    double sum_text, sum_ts;    // C
};
static_assert(sizeof(SlabRec) == 64, "slab record layout");
constexpr int REC_FLOATS = MAX_ROWS * PG * (int)(sizeof(SlabRec) / sizeof(float));

__device__ __forceinline__ float wave_max(float v) { return wave_max_desc(v); }  // (common.hpp: no LDS shuffles, same order)
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_desc(v); }
__device__ __forceinline__ double wave_sum_d(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// (value, log value, index) arg-max, the smallest index winning ties; idx < 0 = no candidate
__device__ __forceinline__ void take_better(float &v, float &l, int &idx, float ov, float ol, int oi) {
    if (oi >= 0 && (idx < 0 || ov > v || (ov == v && oi < idx))) {
        v = ov;
        l = ol;
        idx = oi;
    }
}
__device__ __forceinline__ void wave_argmax(float &v, float &l, int &idx) {
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o), ol = __shfl_xor(l, o);
        const int oi = __shfl_xor(idx, o);
        take_better(v, l, idx, ov, ol, oi);
    }
}

struct RowView {
    const float *lg;
    float *wk;
    SlabRec *rec;  // the row's PG records
    int s0, s1;    // this workgroup's slab
};
__device__ __forceinline__ RowView row_view(const float *logits, long ld, float *work, const LogitsRowState &S, int nv) {
    const int r = blockIdx.y, g = blockIdx.x, sl = (nv + PG - 1) / PG;
    RowView V;
    V.lg = logits + (long)S.logits_row * ld;
    V.wk = work + REC_FLOATS + (long)r * nv;
    V.rec = reinterpret_cast<SlabRec *>(work) + r * PG;
    V.s0 = min(nv, g * sl);
    V.s1 = min(nv, V.s0 + sl);
    return V;
}

// ---- A: rules + maxima ----
__global__ __launch_bounds__(PT) void logits_rules_kernel(const float *__restrict__ logits, long ld,
                                                          const unsigned char *__restrict__ suppress, LogitsParams P,
                                                          const LogitsRowState *__restrict__ rows, float *__restrict__ work) {
    __shared__ float x3[3][PW];
    const int tid = threadIdx.x;
    const LogitsRowState S = rows[blockIdx.y];
    const RowView V = row_view(logits, ld, work, S, P.n_vocab);
    const int nv = P.n_vocab, beg = P.beg;
    const bool initial = S.n_sampled == 0;
    const int ts_lo = S.has_ts ? beg + S.seek_delta / 2 : beg;                       // timestamps below are suppressed
    const int ts_hi = (initial && P.tid0_initial >= 0) ? beg + P.tid0_initial : nv;  // timestamps above are suppressed
    // every load first, unconditional (clamped index): a load under `if (i < s1)` is a branch and a wait per element
    float lv[PV];
    unsigned char sv[PV];
#pragma unroll
    for (int k = 0; k < PV; ++k) {
        const int i = min(V.s0 + tid + k * PT, nv - 1);
        lv[k] = V.lg[i];
        sv[k] = suppress[i];
    }
    float mx = -INFINITY, mx_text = -INFINITY, mx_ts = -INFINITY;
#pragma unroll
    for (int k = 0; k < PV; ++k) {
        const int i = V.s0 + tid + k * PT;
        if (i < V.s1) {
            float v = lv[k];
            if (P.temperature > 0.0f) v /= P.temperature;
            bool kill = sv[k] != 0;
            if (P.suppress_blank && initial && (i == P.eot || i == P.space_id)) kill = true;
            if (S.last_is_ts) {
                if (S.penult_is_ts) kill |= i >= beg;
                else kill |= i < P.eot;
            }
            if (i >= beg && (i < ts_lo || i > ts_hi)) kill = true;
            if (kill) v = -INFINITY;
            V.wk[i] = v;
            mx = fmaxf(mx, v);
            if (i < beg) mx_text = fmaxf(mx_text, v);
            else mx_ts = fmaxf(mx_ts, v);
        }
    }
    mx = wave_max(mx);
    mx_text = wave_max(mx_text);
    mx_ts = wave_max(mx_ts);
    if ((tid & 63) == 0) {
        x3[0][tid >> 6] = mx;
        x3[1][tid >> 6] = mx_text;
        x3[2][tid >> 6] = mx_ts;
    }
    __syncthreads();
    if (tid < 3) {
        float m = x3[tid][0];
        for (int w = 1; w < PW; ++w) m = fmaxf(m, x3[tid][w]);
        float *dst = &V.rec[blockIdx.x].mx;
        dst[tid] = m;
    }
}

// the row's PG records -> LDS with one 16-byte load per thread of wave 0.  Folding them straight from global
// memory compiles to ~40 dependent scalar loads (one round trip each): 8-10 us of a 15 us kernel.
__device__ __forceinline__ const SlabRec *stage_records(const SlabRec *rec, SlabRec *lds_rec) {
    static_assert(PG * sizeof(SlabRec) == 64 * sizeof(uint4), "one uint4 per lane of one wave");
    if (threadIdx.x < 64) reinterpret_cast<uint4 *>(lds_rec)[threadIdx.x] = reinterpret_cast<const uint4 *>(rec)[threadIdx.x];
    __syncthreads();
    return lds_rec;
}

__device__ __forceinline__ float row_max(const SlabRec *rec) {
    float m = rec[0].mx;
    for (int g = 1; g < PG; ++g) m = fmaxf(m, rec[g].mx);
    return m;
}
__device__ __forceinline__ float row_lse(const SlabRec *rec, float mx) {
    float se = 0.0f;
    for (int g = 0; g < PG; ++g) se += rec[g].se;
    return logf(se) + mx;
}

// ---- B: sum exp(v - max) ----
__global__ __launch_bounds__(PT) void logits_sumexp_kernel(const float *__restrict__ logits, long ld, LogitsParams P,
                                                           const LogitsRowState *__restrict__ rows, float *__restrict__ work) {
    __shared__ float part[PW];
    const int tid = threadIdx.x;
    const LogitsRowState S = rows[blockIdx.y];
    const RowView V = row_view(logits, ld, work, S, P.n_vocab);
    float val[PV];
#pragma unroll
    for (int k = 0; k < PV; ++k) {
        const int i = V.s0 + tid + k * PT;
        const float t = V.wk[min(i, P.n_vocab - 1)];  // unconditional load, selected afterwards
        val[k] = i < V.s1 ? t : -INFINITY;
    }
    __shared__ SlabRec srec[PG];
    const float mx = row_max(stage_records(V.rec, srec));
    float se = 0.0f;
#pragma unroll
    for (int k = 0; k < PV; ++k)
        if (val[k] > -INFINITY) se += expf(val[k] - mx);
    se = wave_sum(se);
    if ((tid & 63) == 0) part[tid >> 6] = se;
    __syncthreads();
    if (tid == 0) {
        float m = 0.0f;
        for (int w = 0; w < PW; ++w) m += part[w];
        V.rec[blockIdx.x].se = m;
    }
}

// ---- C: per-class statistics of the slab ----
__global__ __launch_bounds__(PT) void logits_stats_kernel(const float *__restrict__ logits, long ld, LogitsParams P,
                                                          const LogitsRowState *__restrict__ rows, float *__restrict__ work) {
    __shared__ float f_st[PW], f_bp[PW], f_bl[PW], f_tp[PW], f_tl[PW];
    __shared__ int i_b[PW], i_t[PW];
    __shared__ double d_text[PW], d_ts[PW];
    const int tid = threadIdx.x, beg = P.beg;
    const LogitsRowState S = rows[blockIdx.y];
    const RowView V = row_view(logits, ld, work, S, P.n_vocab);
    float val[PV];
#pragma unroll
    for (int k = 0; k < PV; ++k) {
        const int i = V.s0 + tid + k * PT;
        const float t = V.wk[min(i, P.n_vocab - 1)];  // unconditional load, selected afterwards
        val[k] = i < V.s1 ? t : -INFINITY;
    }
    __shared__ SlabRec srec[PG];
    const SlabRec *rec = stage_records(V.rec, srec);
    const float mx = row_max(rec);
    const float lse = row_lse(rec, mx);
    float mx_ts = rec[0].mx_ts;
    for (int g = 1; g < PG; ++g) mx_ts = fmaxf(mx_ts, rec[g].mx_ts);
    const float lp_ts_max = mx_ts > -INFINITY ? mx_ts - lse : -INFINITY;
    float st = 0.0f, best_p = 0.0f, best_lp = 0.0f, best_tp = 0.0f, best_tlp = 0.0f;
    int best_i = -1, best_ti = -1;
    double sum_text = 0.0, sum_ts = 0.0;
#pragma unroll
    for (int k = 0; k < PV; ++k) {
        const int i = V.s0 + tid + k * PT;
        const float v = val[k];
        if (i < V.s1 && v > -INFINITY) {
            const float lp = v - lse;
            const float p = expf(lp);
            if (i >= beg) {
                st += expf(lp - lp_ts_max);
                sum_ts += (double)p;
                if (p > best_tp) {  // indices visited in increasing order per thread: first maximum is kept
                    best_tp = p;
                    best_tlp = lp;
                    best_ti = i;
                }
            } else {
                sum_text += (double)p;
                if (p > best_p) {
                    best_p = p;
                    best_lp = lp;
                    best_i = i;
                }
            }
        }
    }
    st = wave_sum(st);
    sum_text = wave_sum_d(sum_text);
    sum_ts = wave_sum_d(sum_ts);
    wave_argmax(best_p, best_lp, best_i);
    wave_argmax(best_tp, best_tlp, best_ti);
    if ((tid & 63) == 0) {
        const int w = tid >> 6;
        f_st[w] = st;
        f_bp[w] = best_p;
        f_bl[w] = best_lp;
        i_b[w] = best_i;
        f_tp[w] = best_tp;
        f_tl[w] = best_tlp;
        i_t[w] = best_ti;
        d_text[w] = sum_text;
        d_ts[w] = sum_ts;
    }
    __syncthreads();
    if (tid == 0) {
        SlabRec &R = V.rec[blockIdx.x];
        float a = 0.0f;
        double x = 0.0, y = 0.0;
        float bp = 0.0f, bl = 0.0f, tp = 0.0f, tl = 0.0f;
        int bi = -1, ti = -1;
        for (int w = 0; w < PW; ++w) {
            a += f_st[w];
            x += d_text[w];
            y += d_ts[w];
            take_better(bp, bl, bi, f_bp[w], f_bl[w], i_b[w]);
            take_better(tp, tl, ti, f_tp[w], f_tl[w], i_t[w]);
        }
        R.st = a;
        R.best_p = bp;
        R.best_lp = bl;
        R.best_i = bi;
        R.best_tp = tp;
        R.best_tlp = tl;
        R.best_ti = ti;
        R.sum_text = x;
        R.sum_ts = y;
    }
}

// ---- D: the timestamp-mass rule, results, dumps, inverse-CDF draws ----
__global__ __launch_bounds__(PT) void logits_sample_kernel(const float *__restrict__ logits, long ld, LogitsParams P,
                                                           const LogitsRowState *__restrict__ rows,
                                                           LogitsRowResult *__restrict__ res, float *__restrict__ work,
                                                           float *__restrict__ dump_logits, float *__restrict__ dump_logprobs,
                                                           float *__restrict__ dump_probs) {
    __shared__ double wtot[PW];
    const int tid = threadIdx.x, r = blockIdx.y, g = blockIdx.x, nv = P.n_vocab, beg = P.beg;
    const LogitsRowState &S = rows[r];  // by reference: a by-value copy lives in scratch (u[] is indexed in loops)
    const RowView V = row_view(logits, ld, work, S, nv);
    __shared__ SlabRec srec[PG];
    const SlabRec *rec = stage_records(V.rec, srec);
    // fold the row's records in slab order (every workgroup of the row computes the same values)
    const float mx = row_max(rec);
    const float lse = row_lse(rec, mx);
    float mx_text = rec[0].mx_text, mx_ts = rec[0].mx_ts, st = 0.0f;
    float best_p = 0.0f, best_lp = 0.0f, best_tp = 0.0f, best_tlp = 0.0f;
    int best_i = -1, best_ti = -1;
    double sum_ts = 0.0;
    for (int q = 0; q < PG; ++q) {
        mx_text = fmaxf(mx_text, rec[q].mx_text);
        mx_ts = fmaxf(mx_ts, rec[q].mx_ts);
        st += rec[q].st;
        sum_ts += rec[q].sum_ts;
        take_better(best_p, best_lp, best_i, rec[q].best_p, rec[q].best_lp, rec[q].best_i);
        take_better(best_tp, best_tlp, best_ti, rec[q].best_tp, rec[q].best_tlp, rec[q].best_ti);
    }
    const float lp_ts_max = mx_ts > -INFINITY ? mx_ts - lse : -INFINITY;
    const float ts_logprob = st > 0.0f ? logf(st) + lp_ts_max : -INFINITY;
    const float max_text_lp = mx_text > -INFINITY ? mx_text - lse : -INFINITY;
    const bool force_ts = ts_logprob > max_text_lp;  // the timestamps outweigh every text token: text is masked
    LogitsRowResult *out = res + r;
    if (g == 0 && tid == 0) {
        out->tid = best_ti;
        out->pt = (float)((double)best_tp / (sum_ts + 1e-10));
        out->ptsum = (float)sum_ts;
        if (P.n_samples == 0) {
            float bp = force_ts ? 0.0f : best_p, bl = best_lp;
            int bi = force_ts ? -1 : best_i;
            take_better(bp, bl, bi, best_tp, best_tlp, best_ti);
            out->id[0] = bi < 0 ? 0 : bi;
            out->p[0] = bi < 0 ? 0.0f : bp;
            out->plog[0] = bi < 0 ? 0.0f : bl;
        }
    }
    if (dump_logits) {
        for (int i = V.s0 + tid; i < V.s1; i += PT) {
            float v = V.wk[i];
            if (force_ts && i < beg) v = -INFINITY;
            const float lp = v > -INFINITY ? v - lse : -INFINITY;
            dump_logits[(long)r * nv + i] = v;
            dump_logprobs[(long)r * nv + i] = lp;
            dump_probs[(long)r * nv + i] = v == -INFINITY ? 0.0f : expf(lp);
        }
    }
    if (P.n_samples <= 0) return;
    // slab bases of the cumulative distribution (normalised doubles)
    double total = 0.0, below = 0.0, upto = 0.0;
    for (int q = 0; q < PG; ++q) {
        const double t = (force_ts ? 0.0 : rec[q].sum_text) + rec[q].sum_ts;
        if (q == g) below = total;
        total += t;
        if (q == g) upto = total;
    }
    const double base = below / total, next = upto / total;
    if (!(total > 0.0)) {  // nothing left to draw from (cannot happen with EOT alive): defined, not stale, output
        if (g == 0 && tid == 0)
            for (int s = 0; s < P.n_samples; ++s) {
                out->id[s] = 0;
                out->p[s] = 0.0f;
                out->plog[s] = -INFINITY;
            }
        return;
    }
    bool any = false;
    for (int s = 0; s < P.n_samples; ++s) {
        const double u = S.u[s];
        any |= (g == 0 || u > base) && (g == PG - 1 || u <= next);
    }
    if (!any) return;  // uniform over the workgroup
    // thread t owns a contiguous chunk of the slab
    const int sl = V.s1 - V.s0, ch = (sl + PT - 1) / PT;  // <= PV (checked at launch)
    const int i0 = min(V.s1, V.s0 + tid * ch), i1 = min(V.s1, i0 + ch);
    float cv[PV];  // the chunk's masked logits, loaded in one batch (a load per loop iteration is a round trip each)
#pragma unroll
    for (int k = 0; k < PV; ++k) {
        const int i = i0 + k;
        const float t = V.wk[min(i, nv - 1)];
        cv[k] = (i < i1 && !(force_ts && i < beg)) ? t : -INFINITY;
    }
    double pk[PV], local = 0.0;  // normalised probabilities of the chunk: computed once, reused by every draw below
#pragma unroll
    for (int k = 0; k < PV; ++k) {
        pk[k] = cv[k] > -INFINITY ? (double)expf(cv[k] - lse) / total : 0.0;
        if (cv[k] > -INFINITY) local += pk[k];
    }
    double incl = local;
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(incl, o);
        if ((tid & 63) >= o) incl += t;
    }
    if ((tid & 63) == 63) wtot[tid >> 6] = incl;
    __syncthreads();
    double run = base;
    for (int w = 0; w < (tid >> 6); ++w) run += wtot[w];
    // a thread's lower bound is bitwise its predecessor's upper bound: the intervals tile [base, ...) without gaps
    const double prev = __shfl_up(incl, 1);
    const double excl = (tid & 63) == 0 ? run : run + prev;
    incl += run;
    const bool last_owner = (i1 == V.s1) && i0 < V.s1;
    for (int s = 0; s < P.n_samples; ++s) {
        const double u = S.u[s];
        if (!((g == 0 || u > base) && (g == PG - 1 || u <= next))) continue;
        const bool mine = (u > excl && (u <= incl || last_owner)) || (u <= excl && tid == 0);
        if (mine && i0 < i1) {
            double c = excl;
            int pick = -1, lastpos = -1;
            float vp = -INFINITY, vlast = -INFINITY;  // masked logit of the pick / of the last live entry
#pragma unroll
            for (int k = 0; k < PV; ++k) {
                const int i = i0 + k;
                if (i < i1 && pick < 0) {
                    if (cv[k] > -INFINITY) {
                        c += pk[k];
                        lastpos = i;
                        vlast = cv[k];
                    }
                    if (c >= u) {
                        pick = i;
                        vp = cv[k];
                    }
                }
            }
            if (pick < 0) {
                pick = lastpos >= 0 ? lastpos : i1 - 1;
                vp = lastpos >= 0 ? vlast : -INFINITY;
            }
            const float v = vp;
            out->id[s] = pick;
            out->plog[s] = v > -INFINITY ? v - lse : -INFINITY;
            out->p[s] = v > -INFINITY ? expf(v - lse) : 0.0f;
        }
    }
}

}  // namespace

size_t process_logits_work_floats(int n_rows_max, int n_vocab) { return (size_t)REC_FLOATS + (size_t)n_rows_max * n_vocab; }

void launch_process_logits(const float *logits, long ld, const unsigned char *suppress, const LogitsParams &p,
                           const LogitsRowState *d_rows, LogitsRowResult *d_res, int n_rows, float *work,
                           float *dump_logits, float *dump_logprobs, float *dump_probs, hipStream_t st) {
    MW_CHECK(n_rows > 0 && n_rows <= MAX_ROWS && p.n_vocab > 0 && p.beg > 0 && p.beg < p.n_vocab,
             "process_logits: bad vocabulary layout / row count");
    MW_CHECK(p.n_samples >= 0 && p.n_samples <= MAX_CAND, "process_logits: too many samples per row");
    MW_CHECK(ld >= p.n_vocab && work && suppress, "process_logits: buffers");
    MW_CHECK((p.n_vocab + PG - 1) / PG <= PT * PV, "process_logits: vocabulary larger than the per-thread register tile");
    const dim3 grid(PG, n_rows), block(PT);
    hipLaunchKernelGGL(logits_rules_kernel, grid, block, 0, st, logits, ld, suppress, p, d_rows, work);
    hipLaunchKernelGGL(logits_sumexp_kernel, grid, block, 0, st, logits, ld, p, d_rows, work);
    hipLaunchKernelGGL(logits_stats_kernel, grid, block, 0, st, logits, ld, p, d_rows, work);
    const dim3 grid_d((p.n_samples > 0 || dump_logits) ? PG : 1, n_rows);
    hipLaunchKernelGGL(logits_sample_kernel, grid_d, block, 0, st, logits, ld, p, d_rows, d_res, work, dump_logits, dump_logprobs,
                       dump_probs);
    MW_HIP(hipGetLastError());
}

}  // namespace mw
