"""Generates tests/golden/pooler_att.npz by running the reference's own pooler module
(/root/reference/src/tevatron/modeling/mul_grain.py: MULGRAIN_Pooler, 'att' pooling, tied) on seeded inputs.

    PYTHONPATH=/root/reference/src python tests/golden/make_golden_pooler.py

The fixture holds data only: inputs (hidden states, Linear weight / bias) and the module's outputs
(pooled representation, attention weights) for the query and the passage branch. Shapes follow
src/pretrain_finetune1.sh:22-23 (aspect_num=3, aspect tokens from position 1)."""
import os

import numpy as np
import torch
from tevatron.modeling.mul_grain import MULGRAIN_Pooler

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.manual_seed(20240611)
    out = {}
    for tag, (n, L, d, A, begin) in {"small": (7, 9, 64, 3, 1), "bert": (5, 32, 768, 3, 1), "wide": (4, 12, 96, 5, 2)}.items():
        p = MULGRAIN_Pooler(input_dim=d, out_dim=d, q_pool_type="att", doc_pool_type="att", tied=True, aspect_num=A,
                            begin_id=begin)
        with torch.no_grad():
            p.cls_linear_q.weight.mul_(8.0)          # spread the logits so the softmax is not near-uniform
            p.cls_linear_q.bias.uniform_(-1.0, 1.0)
            hq = torch.randn(n, L, d)
            hp = torch.randn(n, L, d)
            q_rep, q_asp, q_att = p(q=hq)
            p_rep, p_asp, p_att = p(doc=hp)
        assert p.cls_linear_d is p.cls_linear_q
        out[f"{tag}_W"] = p.cls_linear_q.weight.detach().numpy().copy()
        out[f"{tag}_b"] = p.cls_linear_q.bias.detach().numpy().copy()
        out[f"{tag}_begin"] = np.int64(begin)
        out[f"{tag}_hq"], out[f"{tag}_hp"] = hq.numpy(), hp.numpy()
        out[f"{tag}_q_rep"], out[f"{tag}_q_att"] = q_rep.numpy(), q_att.numpy().reshape(n, A)
        out[f"{tag}_p_rep"], out[f"{tag}_p_att"] = p_rep.numpy(), p_att.numpy().reshape(n, A)
        assert np.array_equal(q_asp.numpy(), hq.numpy()[:, begin:begin + A])
    np.savez_compressed(os.path.join(HERE, "pooler_att.npz"), **out)
    print("wrote pooler_att.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
