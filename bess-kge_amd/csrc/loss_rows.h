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
template <int KIND, bool ADV, bool GRAD, int CH>
__device__ __forceinline__ void loss_row_impl(const bess_loss_desc& l, const float* __restrict__ pos,
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
        if (lane == 0) row_loss[s] = l.loss_scale * w * (lse - p);
        if (row_norm && lane == 0) {  // (m, L / C) of the row's softmax: see bess_combine_dq_partials
            row_norm[2 * s] = m;
            row_norm[2 * s + 1] = z / (l.loss_scale * w);
        }
        if (GRAD) {
            const float c = l.loss_scale * w;
            if (lane == 0) d_pos[s] = c * (expf(p - lse) - 1.f);
            sweep_grad([&](float x) { return c * expf(x + l.ssce_shift - lse); });
        }
        return;
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
    if (KIND == BESS_LOSS_LOGSIGMOID) {
        if (lane == 0) {
            row_loss[s] = -0.5f * l.loss_scale * w * (log_sigmoid(p + l.margin) + acc);
            if (GRAD) d_pos[s] = -0.5f * l.loss_scale * w * sigmoidf(-(p + l.margin));
        }
    } else {
        if (GRAD) dsum = wave_allreduce_sum(dsum);
        if (lane == 0) {
            row_loss[s] = l.loss_scale * w * acc;
            if (GRAD) d_pos[s] = -dsum;
        }
    }
}

template <int KIND, bool ADV, bool GRAD, int CH>
__device__ __forceinline__ void loss_row(const bess_loss_desc& l, const float* __restrict__ pos,
                                         const float* __restrict__ neg, int64_t s, int64_t n_neg, int64_t ld_neg,
                                         const float* __restrict__ weight, int64_t weight_len, float* __restrict__ row_loss,
                                         float* __restrict__ d_pos, float* __restrict__ d_neg, int64_t ld_dneg,
                                         float* __restrict__ row_norm) {
    loss_row_impl<KIND, ADV, GRAD, CH>(l, pos, neg, s, n_neg, ld_neg, weight, weight_len, row_loss, d_pos, d_neg, ld_dneg,
                                       row_norm);
}


}  // namespace bess
