"""The fp16 path (the arithmetic of BASELINE configs[3], ogbl-wikikg2 TransE fp16) against fixtures made by the
REFERENCE in its fp16 mode: `tests/golden/bess_half.npz` = the reference's `BessKGE.forward` + autograd under
`model.half()` (notebooks/3_wikikg2_fp16.ipynb:300-392; scoring.py:194-197, 342) on torch's CPU half kernels
(`tests/golden/make_golden.py: gen_bess_half`).

The kernels read fp16 rows and accumulate in fp32; TransE / RotatE with p = 1, shared negatives and W % 32 == 0
take the packed-fp16 kernel (`k_l1_fwd_pk`, query rounded to fp16 like the reference's fp16 `h + r`) and its
`ROUND16` backward.  The reference rounds every element-wise result and the final score to fp16 and accumulates
gradients in fp16, so agreement is to fp16 resolution - the tolerances are stated in tests/test_oracle.py
(`HALF_*`) and used unchanged here."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_oracle import (HALF_GRAD_TOL, HALF_LOSS_RTOL, assert_half_scores_close, half_cases,  # noqa: E402
                         load_bess_case)

KEYS = ("head", "relation", "tail", "negative", "negative_mask")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def _model(case, dev, fp32_math):
    from test_hip_parity import build_model

    c = load_bess_case(case)
    c32 = dict(c, table=c["table"].float(), rel=c["rel"].float())
    model = build_model(c32, dev)
    model.score_fn.fp32_math = fp32_math
    return c, model


@pytest.mark.parametrize("fp32_math", [False, True])
@pytest.mark.parametrize("case", half_cases())
def test_forward_matches_the_reference_fp16_mode(dev, case, fp32_math):
    """Scores and loss of every micro-batch and replica; with the packed-fp16 / split kernels (default) and
    with the plain fp32-arithmetic kernels (`fp32_math`) - both are within fp16 resolution of the reference."""
    from besskge import runtime

    c, model = _model(case, dev, fp32_math)
    n, bps = c["meta"]["n_shard"], c["meta"]["bps"]
    bilinear = c["spec"].scorer in ("DistMult", "ComplEx")
    runner = runtime.inference_model(model, runtime.Options(device_iterations=bps), device=dev, dtype=torch.float16)
    res = runner(**{k: c["batch"][k].flatten(end_dim=1) for k in KEYS if k in c["batch"]})
    assert res["positive_score"].dtype == torch.float16  # returned in the table dtype, like the reference
    S = c["outs"]["positive_score"].shape[-1]
    pos = res["positive_score"].float().cpu().reshape(bps, n, S)
    neg = res["negative_score"].float().cpu().reshape(bps, n, S, -1)
    loss = res["loss"].float().cpu().reshape(bps, n)
    for it in range(bps):
        for r in range(n):
            assert_half_scores_close(pos[it, r], c["outs"]["positive_score"][it, r], bilinear)
            assert_half_scores_close(neg[it, r], c["outs"]["negative_score"][it, r], bilinear)
    torch.testing.assert_close(loss, c["outs"]["loss"].float(), rtol=HALF_LOSS_RTOL, atol=1e-3)


@pytest.mark.parametrize("case", half_cases())
def test_sgd_step_matches_the_reference_fp16_gradients(dev, case):
    """One round-once SGD step on the fp16 tables = fp16(table - lr * reference gradient), within the stated
    gradient tolerance (x lr) plus one fp16 ulp of the row value."""
    from besskge import runtime

    c, model = _model(case, dev, False)
    lr = 0.05
    runner = runtime.training_model(model, optimizer=runtime.SGD(lr=lr), device=dev, dtype=torch.float16)
    runner(**{k: c["batch"][k][0] for k in KEYS if k in c["batch"]})
    for got, before, grad in (
            (model.score_fn.entity_embedding, c["table"], c["grads"]["entity"].float()),
            (model.score_fn.relation_embedding, c["rel"], c["grads"]["relation"].float().sum(0))):
        got = got.detach().float().cpu()
        want = before.float() - lr * grad
        ulp = np.float32(2.0) ** (torch.floor(torch.log2(want.abs().clamp(min=2.0 ** -14))) - 10)
        allowed = lr * HALF_GRAD_TOL * float(grad.abs().max()) + ulp
        assert bool(((got - want).abs() <= allowed).all()), float(((got - want).abs() - allowed).max())
        untouched = (grad.reshape(-1, grad.shape[-1]) == 0).all(dim=-1)
        assert torch.equal(got.reshape(-1, grad.shape[-1])[untouched], before.float().reshape(-1, grad.shape[-1])[untouched])
