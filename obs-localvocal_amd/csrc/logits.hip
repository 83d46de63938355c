// logits.hip — K9/K10: fused logits rules + samplers on the device.
//
// Replaces whisper.cpp's whisper_process_logits + whisper_sample_token[_topk], which
// run on the host per decoder per step inside whisper_full
// (/root/reference/src/whisper-utils/whisper-processing.cpp:218-219).  One 1024-thread workgroup per
// decoder row makes four coalesced passes over the n_vocab logits (L2-resident after the first):
//   1. temperature + suppression rules (static mask, blank, timestamp pairing, max_initial_ts,
//      monotonic timestamps) and the row / text / timestamp maxima;
//   2. sum exp -> logsumexp;          3. timestamp logsumexp -> "timestamp mass > best text" rule;
//   4. probabilities: first-max argmax (greedy), timestamp statistics (double), total mass;
//   5. (sampling only) inverse-CDF draws from host-supplied uniforms — the distribution
//      std::discrete_distribution realises in the reference (normalised doubles, lower_bound).
// Only a 100-byte record per row crosses PCIe per step.  Bound: HBM/latency (n_vocab * 4 B per row).
#include "kernels.hpp"

namespace mw {

namespace {

constexpr int NT = 1024;
constexpr int NW = NT / 64;

struct Red {
    float f[NW];
    double d[NW];
    int i[NW];
    float bf;
    double bd;
    int bi;
};

__device__ __forceinline__ float block_max(float v, Red &R) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) R.f[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = R.f[0];
        for (int w = 1; w < NW; ++w) m = fmaxf(m, R.f[w]);
        R.bf = m;
    }
    __syncthreads();
    return R.bf;
}
__device__ __forceinline__ float block_sum(float v, Red &R) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) R.f[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = 0.0f;
        for (int w = 0; w < NW; ++w) m += R.f[w];
        R.bf = m;
    }
    __syncthreads();
    return R.bf;
}
__device__ __forceinline__ double block_sum_d(double v, Red &R) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) R.d[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = 0.0;
        for (int w = 0; w < NW; ++w) m += R.d[w];
        R.bd = m;
    }
    __syncthreads();
    return R.bd;
}
// (value, index) arg-max with the smallest index winning ties; values <= 0 never win (idx stays -1)
__device__ __forceinline__ void block_argmax(float &v, int &idx, Red &R) {
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(idx, o);
        if (oi >= 0 && (idx < 0 || ov > v || (ov == v && oi < idx))) {
            v = ov;
            idx = oi;
        }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        R.f[threadIdx.x >> 6] = v;
        R.i[threadIdx.x >> 6] = idx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bv = 0.0f;
        int bi = -1;
        for (int w = 0; w < NW; ++w)
            if (R.i[w] >= 0 && (bi < 0 || R.f[w] > bv || (R.f[w] == bv && R.i[w] < bi))) {
                bv = R.f[w];
                bi = R.i[w];
            }
        R.bf = bv;
        R.bi = bi;
    }
    __syncthreads();
    v = R.bf;
    idx = R.bi;
}

// three maxima in one synchronisation round (same reduction trees as block_max)
__device__ __forceinline__ void block_max3(float &a, float &b, float &c, Red &R, float (*x3)[NW]) {
    for (int o = 32; o > 0; o >>= 1) {
        a = fmaxf(a, __shfl_xor(a, o));
        b = fmaxf(b, __shfl_xor(b, o));
        c = fmaxf(c, __shfl_xor(c, o));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        x3[0][threadIdx.x >> 6] = a;
        x3[1][threadIdx.x >> 6] = b;
        x3[2][threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float m = x3[threadIdx.x][0];
        for (int w = 1; w < NW; ++w) m = fmaxf(m, x3[threadIdx.x][w]);
        x3[threadIdx.x][0] = m;
    }
    __syncthreads();
    a = x3[0][0];
    b = x3[1][0];
    c = x3[2][0];
}

constexpr int VPT = 51;  // vocabulary entries per thread kept in registers: NT * VPT = 52 224 >= any Whisper vocabulary

__global__ __launch_bounds__(NT) void process_logits_kernel(const float *__restrict__ logits, long ld,
                                                            const unsigned char *__restrict__ suppress, LogitsParams P,
                                                            const LogitsRowState *__restrict__ rows,
                                                            LogitsRowResult *__restrict__ res, float *__restrict__ work,
                                                            float *__restrict__ dump_logits,
                                                            float *__restrict__ dump_logprobs,
                                                            float *__restrict__ dump_probs) {
    __shared__ Red R;
    __shared__ float x3[3][NW];
    __shared__ Red R2, R3, R4;
    __shared__ double scan[NT];
    const int r = blockIdx.x, tid = threadIdx.x;
    const LogitsRowState S = rows[r];
    const float *lg = logits + (long)S.logits_row * ld;
    float *wk = work + (long)r * P.n_vocab;
    const int nv = P.n_vocab, beg = P.beg;
    const bool initial = S.n_sampled == 0;
    const int ts_lo = S.has_ts ? beg + S.seek_delta / 2 : beg;               // timestamps below are suppressed
    const int ts_hi = (initial && P.tid0_initial >= 0) ? beg + P.tid0_initial : nv;  // timestamps above are suppressed

    // pass 1: rules + maxima.  Thread t owns entries t, t + NT, ...; they stay in registers for passes 2-4
    // (the masked row is also written to `work` for the chunk-ordered inverse-CDF pass)
    float val[VPT];
    float mx = -INFINITY, mx_text = -INFINITY, mx_ts = -INFINITY;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int i = tid + k * NT;
        float v = -INFINITY;
        if (i < nv) {
            v = lg[i];
            if (P.temperature > 0.0f) v /= P.temperature;
            bool kill = suppress[i] != 0;
            if (P.suppress_blank && initial && (i == P.eot || i == P.space_id)) kill = true;
            if (S.last_is_ts) {
                if (S.penult_is_ts) kill |= i >= beg;
                else kill |= i < P.eot;
            }
            if (i >= beg && (i < ts_lo || i > ts_hi)) kill = true;
            if (kill) v = -INFINITY;
            wk[i] = v;
            mx = fmaxf(mx, v);
            if (i < beg) mx_text = fmaxf(mx_text, v);
            else mx_ts = fmaxf(mx_ts, v);
        }
        val[k] = v;
    }
    block_max3(mx, mx_text, mx_ts, R, x3);
    // pass 2: logsumexp
    float se = 0.0f;
#pragma unroll
    for (int k = 0; k < VPT; ++k)
        if (val[k] > -INFINITY) se += expf(val[k] - mx);
    const float lse = logf(block_sum(se, R)) + mx;
    // pass 3: logsumexp over timestamps of the log-probabilities
    const float lp_ts_max = mx_ts > -INFINITY ? mx_ts - lse : -INFINITY;
    float st = 0.0f;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int i = tid + k * NT;
        if (i >= beg && val[k] > -INFINITY) st += expf((val[k] - lse) - lp_ts_max);
    }
    st = block_sum(st, R);
    const float ts_logprob = st > 0.0f ? logf(st) + lp_ts_max : -INFINITY;
    const float max_text_lp = mx_text > -INFINITY ? mx_text - lse : -INFINITY;
    const bool force_ts = ts_logprob > max_text_lp;
    // pass 4: probabilities, greedy arg-max, timestamp statistics, total mass
    float best_p = 0.0f;
    int best_i = -1;
    float best_tp = 0.0f;
    int best_ti = -1;
    double sum_ts = 0.0, total = 0.0;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int i = tid + k * NT;
        if (i < nv) {
            float v = val[k];
            if (force_ts && i < beg) {
                v = -INFINITY;
                wk[i] = v;
            }
            const float lp = v > -INFINITY ? v - lse : -INFINITY;
            const float p = v == -INFINITY ? 0.0f : expf(lp);
            if (dump_logits) {
                dump_logits[(long)r * nv + i] = v;
                dump_logprobs[(long)r * nv + i] = lp;
                dump_probs[(long)r * nv + i] = p;
            }
            if (p > best_p) {  // indices visited in increasing order per thread: first maximum is kept
                best_p = p;
                best_i = i;
            }
            total += (double)p;
            if (i >= beg) {
                sum_ts += (double)p;
                if (p > best_tp) {
                    best_tp = p;
                    best_ti = i;
                }
            }
        }
    }
    // the four block reductions share their synchronisation rounds (each keeps its own reduction order)
    {
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best_p, o);
            const int oi = __shfl_xor(best_i, o);
            if (oi >= 0 && (best_i < 0 || ov > best_p || (ov == best_p && oi < best_i))) {
                best_p = ov;
                best_i = oi;
            }
            const float tv = __shfl_xor(best_tp, o);
            const int ti = __shfl_xor(best_ti, o);
            if (ti >= 0 && (best_ti < 0 || tv > best_tp || (tv == best_tp && ti < best_ti))) {
                best_tp = tv;
                best_ti = ti;
            }
            sum_ts += __shfl_xor(sum_ts, o);
            total += __shfl_xor(total, o);
        }
        __syncthreads();
        if ((tid & 63) == 0) {
            const int w = tid >> 6;
            R.f[w] = best_p;
            R.i[w] = best_i;
            R2.f[w] = best_tp;
            R2.i[w] = best_ti;
            R3.d[w] = sum_ts;
            R4.d[w] = total;
        }
        __syncthreads();
        if (tid == 0 || tid == 64) {
            Red &Q = tid == 0 ? R : R2;
            float bv = 0.0f;
            int bi = -1;
            for (int w = 0; w < NW; ++w)
                if (Q.i[w] >= 0 && (bi < 0 || Q.f[w] > bv || (Q.f[w] == bv && Q.i[w] < bi))) {
                    bv = Q.f[w];
                    bi = Q.i[w];
                }
            Q.bf = bv;
            Q.bi = bi;
        } else if (tid == 128 || tid == 192) {
            Red &Q = tid == 128 ? R3 : R4;
            double m = 0.0;
            for (int w = 0; w < NW; ++w) m += Q.d[w];
            Q.bd = m;
        }
        __syncthreads();
        best_p = R.bf;
        best_i = R.bi;
        best_tp = R2.bf;
        best_ti = R2.bi;
        sum_ts = R3.bd;
        total = R4.bd;
    }
    LogitsRowResult *out = res + r;
    if (tid == 0) {
        out->tid = best_ti;
        out->pt = (float)((double)best_tp / (sum_ts + 1e-10));
        out->ptsum = (float)sum_ts;
        if (P.n_samples == 0) {
            const int id = best_i < 0 ? 0 : best_i;
            out->id[0] = id;
            out->p[0] = best_i < 0 ? 0.0f : best_p;
            out->plog[0] = best_i < 0 ? 0.0f : wk[id] - lse;
        }
    }
    if (P.n_samples > 0) {
        // pass 5: inverse CDF over normalised double probabilities; thread t owns a contiguous chunk
        const int ch = (nv + NT - 1) / NT;
        const int i0 = tid * ch, i1 = min(nv, i0 + ch);
        double local = 0.0;
        for (int i = i0; i < i1; ++i) {
            const float v = wk[i];
            if (v > -INFINITY) local += (double)expf(v - lse) / total;
        }
        scan[tid] = local;
        __syncthreads();
        // inclusive scan of 1024 doubles: 64-wide shuffle scan per wave + serial pass over wave totals
        double incl = local;
        for (int o = 1; o < 64; o <<= 1) {
            const double t = __shfl_up(incl, o);
            if ((tid & 63) >= o) incl += t;
        }
        if ((tid & 63) == 63) R.d[tid >> 6] = incl;
        __syncthreads();
        if (tid == 0) {
            double run = 0.0;
            for (int w = 0; w < NW; ++w) {
                const double t = R.d[w];
                R.d[w] = run;
                run += t;
            }
        }
        __syncthreads();
        incl += R.d[tid >> 6];
        const double excl = incl - local;
        const bool last_owner = (i1 == nv) && i0 < nv;
        for (int s = 0; s < P.n_samples; ++s) {
            const double u = S.u[s];
            const bool mine = (u > excl && (u <= incl || last_owner)) || (u <= 0.0 && tid == 0);
            if (mine && i0 < i1) {
                double c = excl;
                int pick = -1, lastpos = -1;
                for (int i = i0; i < i1; ++i) {
                    const float v = wk[i];
                    if (v > -INFINITY) {
                        c += (double)expf(v - lse) / total;
                        lastpos = i;
                    }
                    if (c >= u) {
                        pick = i;
                        break;
                    }
                }
                if (pick < 0) pick = lastpos >= 0 ? lastpos : i1 - 1;
                const float v = wk[pick];
                out->id[s] = pick;
                out->plog[s] = v > -INFINITY ? v - lse : -INFINITY;
                out->p[s] = v > -INFINITY ? expf(v - lse) : 0.0f;
            }
        }
    }
}

}  // namespace

void launch_process_logits(const float *logits, long ld, const unsigned char *suppress, const LogitsParams &p,
                           const LogitsRowState *d_rows, LogitsRowResult *d_res, int n_rows, float *work,
                           float *dump_logits, float *dump_logprobs, float *dump_probs, hipStream_t st) {
    MW_CHECK(n_rows > 0 && p.n_vocab > 0 && p.beg > 0 && p.beg < p.n_vocab, "process_logits: bad vocabulary layout");
    MW_CHECK(p.n_samples >= 0 && p.n_samples <= MAX_CAND, "process_logits: too many samples per row");
    MW_CHECK(ld >= p.n_vocab && work && suppress, "process_logits: buffers");
    MW_CHECK(p.n_vocab <= NT * VPT, "process_logits: vocabulary larger than the per-thread register tile");
    hipLaunchKernelGGL(process_logits_kernel, dim3(n_rows), dim3(NT), 0, st, logits, ld, suppress, p, d_rows, d_res, work,
                       dump_logits, dump_logprobs, dump_probs);
    MW_HIP(hipGetLastError());
}

}  // namespace mw
