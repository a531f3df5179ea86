"""hf_attention.legacy_quant_attention_class on toy attention classes (CPU, no reference, no transformers): which product of
a class's own source takes which KV-cache quantizer -- the reference's rule (plugins/attention.py:45-210): of two
matmul / bmm calls the first found takes v on its second operand and the second q / k (between transposes); of two `@` the
second found takes q / k and the first v; one scaled_dot_product_attention call takes all three; anything else is refused."""

import torch
import torch.nn.functional as F
from torch import nn

import _moa_import

moa = _moa_import.load()
legacy = moa.hf_attention.legacy_quant_attention_class


class TwoMatmulAttention(nn.Module):
    def forward(self, q, k, v):
        scores = torch.matmul(q, k.transpose(-1, -2)) / q.shape[-1] ** 0.5
        return torch.matmul(torch.softmax(scores, dim=-1), v)


class OperatorAttention(nn.Module):
    def forward(self, q, k, v):
        scores = (q @ k.transpose(-1, -2)) / q.shape[-1] ** 0.5
        return torch.softmax(scores, dim=-1) @ v


class FusedAttention(nn.Module):
    def forward(self, q, k, v):
        return F.scaled_dot_product_attention(q, k, v)


class ThreeProductsAttention(nn.Module):
    def forward(self, q, k, v):
        return torch.matmul(torch.matmul(torch.matmul(q, k.transpose(-1, -2)), v), v.transpose(-1, -2))


class _Spy(nn.Module):
    """Stands where a TensorQuantizer would: remembers what it was handed and hands it back."""

    def __init__(self):
        super().__init__()
        self.seen = []

    def forward(self, x):
        self.seen.append(x.detach().clone())
        return x


def _run(cls):
    qcls = legacy(cls)
    assert qcls is not None and issubclass(qcls, cls) and qcls.__name__ == "Quant" + cls.__name__
    m = qcls()
    m.q_bmm_quantizer, m.k_bmm_quantizer, m.v_bmm_quantizer = _Spy(), _Spy(), _Spy()
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(2, 3, 5, 8, generator=g) for _ in range(3))
    out = m(q, k, v)
    assert torch.allclose(out, cls()(q, k, v))  # pass-through quantizers change nothing
    return m, q, k, v


def test_two_matmul_calls_first_found_takes_v_second_takes_q_and_k_between_transposes():
    m, q, k, v = _run(TwoMatmulAttention)
    # breadth-first, the statement-level product (P @ V) is found before the one nested inside the division (Q @ K^T)
    assert torch.equal(m.v_bmm_quantizer.seen[0], v)
    assert torch.equal(m.q_bmm_quantizer.seen[0], q.transpose(-1, -2))  # q too goes through the transpose pair
    assert torch.equal(m.k_bmm_quantizer.seen[0], k)  # (k^T)^T: the key states in [.., tokens, dim] layout, per token


def test_two_matmul_operators_second_found_takes_q_and_k_first_takes_v():
    m, q, k, v = _run(OperatorAttention)
    assert torch.equal(m.v_bmm_quantizer.seen[0], v) and torch.equal(m.q_bmm_quantizer.seen[0], q)
    assert torch.equal(m.k_bmm_quantizer.seen[0], k)


def test_one_fused_attention_call_takes_all_three():
    m, q, k, v = _run(FusedAttention)
    assert all(torch.equal(spy.seen[0], t) for spy, t in ((m.q_bmm_quantizer, q), (m.k_bmm_quantizer, k), (m.v_bmm_quantizer, v)))


def test_a_class_with_another_number_of_products_is_refused_and_the_answer_is_remembered():
    assert legacy(ThreeProductsAttention) is None and legacy(ThreeProductsAttention) is None
    assert legacy(TwoMatmulAttention) is legacy(TwoMatmulAttention)
