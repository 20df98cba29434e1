"""Build-authored torch (CPU, autograd) restatement of the OPNet training step.

TEST INFRASTRUCTURE ONLY.  Used where the checker needs gradients: forward exactly as
oracle/opnet_oracle.py (reference baselines/learned_models.py:35-52) but written with differentiable
torch ops as an explicit time loop; the loss and optimiser restate reference
baselines/training_main.py:150-152,192-217: nn.L1Loss(reduction="none") -> mean, torch.optim.Adam
(lr 1e-3, betas (0.9, 0.999), eps 1e-8, no weight decay).  Pinned against gradients / Adam steps
produced by the reference's own model under torch autograd (tests/golden/opnet_train_*.npz).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch


def lstm_seq(x: torch.Tensor, w_ih: torch.Tensor, w_hh: torch.Tensor) -> torch.Tensor:
    """bias-free single-layer LSTM, gates i,f,g,o, zero initial state; x [B,T,I] -> [B,T,H]"""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    gx = x @ w_ih.t()
    outs = []
    for t in range(T):
        g = gx[:, t] + h @ w_hh.t()
        i, f, gg, o = g.split(H, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, dim=1)


def opnet_forward(boxes: torch.Tensor, p: Dict[str, torch.Tensor]):
    B, T = boxes.shape[:2]
    h1 = lstm_seq(boxes.reshape(B, T, -1), p["object_to_track_LSTM.weight_ih_l0"], p["object_to_track_LSTM.weight_hh_l0"])
    logits = h1 @ p["object_to_track_prediction.weight"].t()
    probs = torch.softmax(logits, dim=-1)
    fb = (boxes * probs.unsqueeze(-1)).sum(dim=2)          # einsum "bfot,bfo->bft"
    h2 = lstm_seq(fb, p["video_LSTM.weight_ih_l0"], p["video_LSTM.weight_hh_l0"])
    y = h2 @ p["prediction_layer.weight"].t()
    return y, logits.permute(0, 2, 1).contiguous()


class OPNetTorch(torch.nn.Module):
    """the same graph on torch's own CPU LSTM op (oneDNN / native) - what the reference's nn.LSTM calls run on
    (learned_models.py:29,32); used as the timed CPU baseline of the TRAINING step, where the explicit python time loop
    of `lstm_seq` above would measure the interpreter rather than the arithmetic"""

    def __init__(self, params: Dict[str, np.ndarray]):
        super().__init__()
        h1 = params["object_to_track_LSTM.weight_hh_l0"].shape[1]
        h2 = params["video_LSTM.weight_hh_l0"].shape[1]
        self.lstm1 = torch.nn.LSTM(90, h1, batch_first=True, bias=False)
        self.lstm2 = torch.nn.LSTM(6, h2, batch_first=True, bias=False)
        self.sel = torch.nn.Linear(h1, 15, bias=False)
        self.out = torch.nn.Linear(h2, 4, bias=False)
        with torch.no_grad():
            self.lstm1.weight_ih_l0.copy_(torch.from_numpy(params["object_to_track_LSTM.weight_ih_l0"]))
            self.lstm1.weight_hh_l0.copy_(torch.from_numpy(params["object_to_track_LSTM.weight_hh_l0"]))
            self.lstm2.weight_ih_l0.copy_(torch.from_numpy(params["video_LSTM.weight_ih_l0"]))
            self.lstm2.weight_hh_l0.copy_(torch.from_numpy(params["video_LSTM.weight_hh_l0"]))
            self.sel.weight.copy_(torch.from_numpy(params["object_to_track_prediction.weight"]))
            self.out.weight.copy_(torch.from_numpy(params["prediction_layer.weight"]))

    def forward(self, boxes: torch.Tensor) -> torch.Tensor:
        B, T = boxes.shape[:2]
        h1, _ = self.lstm1(boxes.reshape(B, T, 90))
        probs = torch.softmax(self.sel(h1), dim=-1)
        frames_boxes = torch.einsum("bfot,bfo->bft", boxes, probs)
        h2, _ = self.lstm2(frames_boxes)
        return self.out(h2)


def l1_mean(y: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """training_main.py:152,192,204 for the supervised models: mean(|y - label|)."""
    return (y - labels).abs().mean()


def loss_and_grads(boxes: np.ndarray, labels: np.ndarray, params: Dict[str, np.ndarray], dtype=torch.float32):
    p = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in params.items()}
    y, _ = opnet_forward(torch.tensor(boxes, dtype=dtype), p)
    loss = l1_mean(y, torch.tensor(labels, dtype=dtype))
    loss.backward()
    return float(loss.item()), {k: v.grad.numpy() for k, v in p.items()}, y.detach().numpy()


def adam_step(params: Dict[str, np.ndarray], grads: Dict[str, np.ndarray], state: dict, lr=1e-3,
              b1=0.9, b2=0.999, eps=1e-8) -> None:
    """torch.optim.Adam.step() (single-tensor formulation), in place, fp32."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    for k in params:
        g = grads[k].astype(np.float32)
        m = state.setdefault("m_" + k, np.zeros_like(g))
        v = state.setdefault("v_" + k, np.zeros_like(g))
        m *= np.float32(b1); m += np.float32(1.0 - b1) * g
        v *= np.float32(b2); v += np.float32(1.0 - b2) * g * g
        denom = np.sqrt(v) / np.float32(np.sqrt(bc2)) + np.float32(eps)
        params[k] -= np.float32(lr / bc1) * (m / denom)


def baseline_lstm_forward(x: torch.Tensor, p: Dict[str, torch.Tensor]) -> torch.Tensor:
    """BaselineLstm.forward (learned_models.py:104-118)"""
    B, T = x.shape[:2]
    h = lstm_seq(x.reshape(B, T, -1), p["video_LSTM.weight_ih_l0"], p["video_LSTM.weight_hh_l0"])
    return h @ p["predictions_layer.weight"].t()


def non_linear_lstm_forward(x: torch.Tensor, p: Dict[str, torch.Tensor]) -> torch.Tensor:
    """NonLinearLstm.forward (learned_models.py:134-151)"""
    B, T = x.shape[:2]
    h = torch.relu(x @ p["boxes_linear.weight"].t()).reshape(B, T, -1)
    for l in range(2):
        h = lstm_seq(h, p[f"video_LSTM.weight_ih_l{l}"], p[f"video_LSTM.weight_hh_l{l}"])
    return h @ p["predictions_layer.weight"].t()


def opnet_lstm_mlp_forward(x: torch.Tensor, p: Dict[str, torch.Tensor]) -> torch.Tensor:
    """OPNetLstmMlp.forward (learned_models.py:72-89): OPNet's selection stage, then relu(Linear 6->H2) -> Linear"""
    B, T = x.shape[:2]
    h1 = lstm_seq(x.reshape(B, T, -1), p["object_to_track_LSTM.weight_ih_l0"], p["object_to_track_LSTM.weight_hh_l0"])
    probs = torch.softmax(h1 @ p["object_to_track_prediction.weight"].t(), dim=-1)
    frames_boxes = torch.einsum("bfot,bfo->bft", x, probs)
    return torch.relu(frames_boxes @ p["hidden_layer.weight"].t()) @ p["prediction_layer.weight"].t()


def transformer_lstm_forward(x: torch.Tensor, p: Dict[str, torch.Tensor], nhead: int) -> torch.Tensor:
    """TransformerLstm.forward (learned_models.py:175-197) without dropout, slot 0 only (the only slot that reaches
    the output; attention runs over the S = B*T tokens of the whole minibatch, as the reference's layout makes it)"""
    B, T = x.shape[:2]
    z = torch.relu(x[:, :, 0, :] @ p["boxes_linear.weight"].t()).reshape(B * T, -1)
    E = z.shape[1]
    hd = E // nhead
    li = 0
    while f"attention_encoder.layers.{li}.linear1.weight" in p:
        pre = f"attention_encoder.layers.{li}."
        qkv = z @ p[pre + "self_attn.in_proj_weight"].t() + p[pre + "self_attn.in_proj_bias"]
        q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
        heads = []
        for h in range(nhead):
            sl = slice(h * hd, (h + 1) * hd)
            heads.append(torch.softmax((q[:, sl] * hd ** -0.5) @ k[:, sl].t(), dim=-1) @ v[:, sl])
        a = torch.cat(heads, dim=1) @ p[pre + "self_attn.out_proj.weight"].t() + p[pre + "self_attn.out_proj.bias"]
        z = torch.nn.functional.layer_norm(z + a, (E,), p[pre + "norm1.weight"], p[pre + "norm1.bias"])
        f = torch.relu(z @ p[pre + "linear1.weight"].t() + p[pre + "linear1.bias"])
        f = f @ p[pre + "linear2.weight"].t() + p[pre + "linear2.bias"]
        z = torch.nn.functional.layer_norm(z + f, (E,), p[pre + "norm2.weight"], p[pre + "norm2.bias"])
        li += 1
    h = z.reshape(B, T, E)
    l = 0
    while f"video_LSTM.weight_ih_l{l}" in p:
        h = lstm_seq(h, p[f"video_LSTM.weight_ih_l{l}"], p[f"video_LSTM.weight_hh_l{l}"])
        l += 1
    return h @ p["predictions_layer.weight"].t()


def sibling_loss_and_grads(name: str, x: np.ndarray, labels: np.ndarray, params: Dict[str, np.ndarray], dtype=torch.float32,
                           nhead: int = 2):
    fwd = {"baseline_lstm": baseline_lstm_forward, "non_linear_lstm": non_linear_lstm_forward,
           "opnet_lstm_mlp": opnet_lstm_mlp_forward,
           "transformer_lstm": lambda xx, pp: transformer_lstm_forward(xx, pp, nhead)}[name]
    p = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in params.items()}
    y = fwd(torch.tensor(x, dtype=dtype), p)
    loss = l1_mean(y, torch.tensor(labels, dtype=dtype))
    loss.backward()
    return float(loss.item()), {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape, v.detach().numpy().dtype))
                                for k, v in p.items()}, y.detach().numpy()
