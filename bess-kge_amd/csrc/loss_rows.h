// K8 row by row: the loss term of one triple and the gradient of its scores, by one wave (loss.hip).
//   reference loss.py:28-51 (self-adversarial weights = detached softmax), 115-134 (log-sigmoid), 179-195 (margin
//   ranking), 224-251 (sampled-softmax cross entropy); fp32, summed over the micro-batch (bess.py:254-260).
#pragma once
#include "common.h"

namespace bess {

__device__ __forceinline__ float log_sigmoid(float x) {
    // = -softplus(-x), the stable form torch uses
    return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// One wave per triple.  CH > 0: the row of negative scores is read ONCE, as CH 16-byte chunks per lane held in
// registers (rows of up to 256 * CH scores, n_neg % 4 == 0, 16-byte aligned rows), and the gradient row is written
// with 16-byte stores; CH = 0: any shape, the row is streamed from memory (cache) in each of the passes, 4 bytes per
// lane; CH = -1: the same with 16-byte accesses (longer rows of the 16-byte-aligned shapes).
// Returns d loss / d positive score (what d_pos[s] gets; in every lane - the fused S-row tail hands it on in registers).
template <int KIND, bool ADV, bool GRAD, int CH>
__device__ __forceinline__ float loss_row_impl(const bess_loss_desc& l, const float* __restrict__ pos,
                                              const float* __restrict__ neg, int64_t s, int64_t n_neg, int64_t ld_neg,
                                              const float* __restrict__ weight, int64_t weight_len,
                                              float* __restrict__ row_loss, float* __restrict__ d_pos,
                                              float* __restrict__ d_neg, int64_t ld_dneg, float* __restrict__ row_norm) {
    const int lane = threadIdx.x & 63;
    const float* nr = neg + s * ld_neg;
    float* dn = GRAD ? d_neg + s * ld_dneg : nullptr;
    const float w = weight[weight_len == 1 ? 0 : s];
    const float p = pos[s];
    const int n = static_cast<int>(n_neg);

    float4 v[CH > 0 ? CH : 1];
    if (CH > 0) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int j = (c * 64 + lane) * 4;
            v[c] = j < n ? *reinterpret_cast<const float4*>(nr + j) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // f(score) for every score of the row
    auto sweep = [&](auto&& f) {
        if (CH > 0) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if ((c * 64 + lane) * 4 < n) {
                    f(v[c].x);
                    f(v[c].y);
                    f(v[c].z);
                    f(v[c].w);
                }
            }
        } else if (CH < 0) {  // streamed, 16 bytes per lane
            for (int j = lane * 4; j < n; j += 256) {
                const float4 x = *reinterpret_cast<const float4*>(nr + j);
                f(x.x);
                f(x.y);
                f(x.z);
                f(x.w);
            }
        } else {
            for (int j = lane; j < n; j += 64) f(nr[j]);
        }
    };
    // d_neg[j] = f(score[j])
    auto sweep_grad = [&](auto&& f) {
        if (CH > 0) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int j = (c * 64 + lane) * 4;
                if (j < n) *reinterpret_cast<float4*>(dn + j) = make_float4(f(v[c].x), f(v[c].y), f(v[c].z), f(v[c].w));
            }
        } else if (CH < 0) {
            for (int j = lane * 4; j < n; j += 256) {
                const float4 x = *reinterpret_cast<const float4*>(nr + j);
                *reinterpret_cast<float4*>(dn + j) = make_float4(f(x.x), f(x.y), f(x.z), f(x.w));
            }
        } else {
            for (int j = lane; j < n; j += 64) dn[j] = f(nr[j]);
        }
    };

    if (KIND == BESS_LOSS_SSCE) {
        // cross entropy of [pos, neg + shift] against class 0
        float m = p;
        sweep([&](float x) { m = fmaxf(m, x + l.ssce_shift); });
        m = wave_allreduce_max(m);
        float z = 0.f;
        sweep([&](float x) { z += expf(x + l.ssce_shift - m); });
        z = wave_allreduce_sum(z) + expf(p - m);
        const float lse = m + logf(z);
        if (lane == 0) store_agent(row_loss + s, l.loss_scale * w * (lse - p));  // (read by the launch's last workgroup)
        if (row_norm && lane == 0) {  // (m, L / C) of the row's softmax: see bess_combine_dq_partials
            row_norm[2 * s] = m;
            row_norm[2 * s + 1] = z / (l.loss_scale * w);
        }
        float gp = 0.f;
        if (GRAD) {
            const float c = l.loss_scale * w;
            gp = c * (expf(p - lse) - 1.f);
            if (lane == 0) d_pos[s] = gp;
            sweep_grad([&](float x) { return c * expf(x + l.ssce_shift - lse); });
        }
        return gp;
    }

    // negative weights: softmax(adversarial_scale * neg) (detached) or 1/N
    float m = 0.f, inv_z = 1.f / static_cast<float>(n);
    if (ADV) {
        m = -INFINITY;
        sweep([&](float x) { m = fmaxf(m, l.adversarial_scale * x); });
        m = wave_allreduce_max(m);
        float z = 0.f;
        sweep([&](float x) { z += expf(l.adversarial_scale * x - m); });
        inv_z = 1.f / wave_allreduce_sum(z);
    }
    if (row_norm && lane == 0) {
        row_norm[2 * s] = m;
        row_norm[2 * s + 1] = (1.f / inv_z) / ((KIND == BESS_LOSS_LOGSIGMOID ? 0.5f : 1.f) * l.loss_scale * w);
    }
    float acc = 0.f, dsum = 0.f;
    auto aw_of = [&](float x) { return ADV ? expf(l.adversarial_scale * x - m) * inv_z : inv_z; };
    if (KIND == BESS_LOSS_LOGSIGMOID) {
        sweep([&](float x) { acc += aw_of(x) * log_sigmoid(-x - l.margin); });
        if (GRAD) sweep_grad([&](float x) { return 0.5f * l.loss_scale * w * aw_of(x) * sigmoidf(x + l.margin); });
    } else {
        sweep([&](float x) { acc += aw_of(x) * fmaxf(x - p + l.margin, 0.f); });
        if (GRAD)
            sweep_grad([&](float x) {
                const float gj = (x - p + l.margin > 0.f) ? l.loss_scale * w * aw_of(x) : 0.f;
                dsum += gj;
                return gj;
            });
    }
    acc = wave_allreduce_sum(acc);
    float gp = 0.f;
    if (KIND == BESS_LOSS_LOGSIGMOID) {
        if (GRAD) gp = -0.5f * l.loss_scale * w * sigmoidf(-(p + l.margin));
        if (lane == 0) {
            store_agent(row_loss + s, -0.5f * l.loss_scale * w * (log_sigmoid(p + l.margin) + acc));
            if (GRAD) d_pos[s] = gp;
        }
    } else {
        if (GRAD) {
            dsum = wave_allreduce_sum(dsum);
            gp = -dsum;
        }
        if (lane == 0) {
            store_agent(row_loss + s, l.loss_scale * w * acc);
            if (GRAD) d_pos[s] = gp;
        }
    }
    return gp;
}

template <int KIND, bool ADV, bool GRAD, int CH>
__device__ __forceinline__ float loss_row(const bess_loss_desc& l, const float* __restrict__ pos,
                                         const float* __restrict__ neg, int64_t s, int64_t n_neg, int64_t ld_neg,
                                         const float* __restrict__ weight, int64_t weight_len, float* __restrict__ row_loss,
                                         float* __restrict__ d_pos, float* __restrict__ d_neg, int64_t ld_dneg,
                                         float* __restrict__ row_norm) {
    return loss_row_impl<KIND, ADV, GRAD, CH>(l, pos, neg, s, n_neg, ld_neg, weight, weight_len, row_loss, d_pos, d_neg,
                                              ld_dneg, row_norm);
}

// d loss / d query of one query from the partials (m_i, l_i, acc_i[W]) its work items left in the fused training
// forward (neg_pertriple.hip: FuseArgs), by one wave:
//   d_query[q, :] = C_q / L_q * sum_items exp(m_i - m) acc_i,  m = max_i m_i (and the positive for ssce),
//   L_q = sum_i exp(m_i - m) l_i (+ exp(pos - m) for ssce);  C_q = loss_scale * w_q (x 1/2 for the log-sigmoid loss).
// With `norm` ([n_query, 2] = (m, L / C_q) taken over ALL the negatives of the query, of which these items hold a
// part - the other parts were scored on other shards, ScoreMoving) the items are only rescaled.
// The row goes to `dq` and, if given, to `dq2` too (the fused tail keeps it in LDS and may also store it).
__device__ __forceinline__ void combine_dq_row(const float* __restrict__ st_ml, const float* __restrict__ st_acc, int64_t q,
                                               int items, int W, int kind, float loss_scale,
                                               const float* __restrict__ pos, const float* __restrict__ weight,
                                               int64_t weight_len, const float* __restrict__ norm, float* dq, float* dq2) {
    const int lane = threadIdx.x & 63;
    float m, scale;
    if (norm) {
        m = norm[2 * q];
        const float l_over_c = norm[2 * q + 1];
        scale = (l_over_c > 0.f && l_over_c < INFINITY) ? 1.f / l_over_c : 0.f;
    } else {
        // (m_i, l_i) of the items, strided over the lanes
        m = -INFINITY;
        for (int i = lane; i < items; i += 64) m = fmaxf(m, st_ml[(q * items + i) * 2]);
        m = wave_allreduce_max(m);
        if (kind == BESS_LOSS_SSCE) m = fmaxf(m, pos[q]);
        float L = 0.f;
        for (int i = lane; i < items; i += 64) {
            const float mi = st_ml[(q * items + i) * 2];
            if (mi != -INFINITY) L += st_ml[(q * items + i) * 2 + 1] * expf(mi - m);
        }
        L = wave_allreduce_sum(L);
        if (kind == BESS_LOSS_SSCE) L += expf(pos[q] - m);
        const float w = weight[weight_len == 1 ? 0 : q];
        const float C = (kind == BESS_LOSS_LOGSIGMOID ? 0.5f : 1.f) * loss_scale * w;
        scale = L > 0.f ? C / L : 0.f;
    }
    const float* ap = st_acc + q * items * W;
    if ((W & 3) == 0) {
        for (int c = lane * 4; c < W; c += 256) {
            float x[4] = {0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < items; ++i) {
                const float mi = st_ml[(q * items + i) * 2];  // wave-uniform, cached
                if (mi == -INFINITY) continue;
                const float f = expf(mi - m);
                const float4 v = *reinterpret_cast<const float4*>(ap + static_cast<int64_t>(i) * W + c);
                x[0] = fmaf(v.x, f, x[0]); x[1] = fmaf(v.y, f, x[1]); x[2] = fmaf(v.z, f, x[2]); x[3] = fmaf(v.w, f, x[3]);
            }
            const float4 o = make_float4(scale * x[0], scale * x[1], scale * x[2], scale * x[3]);
            *reinterpret_cast<float4*>(dq + c) = o;
            if (dq2) *reinterpret_cast<float4*>(dq2 + c) = o;
        }
    } else {
        for (int c = lane; c < W; c += 64) {
            float x = 0.f;
            for (int i = 0; i < items; ++i) {
                const float mi = st_ml[(q * items + i) * 2];
                if (mi != -INFINITY) x = fmaf(ap[static_cast<int64_t>(i) * W + c], expf(mi - m), x);
            }
            dq[c] = scale * x;
            if (dq2) dq2[c] = scale * x;
        }
    }
}


}  // namespace bess
