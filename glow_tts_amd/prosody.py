"""GST prosody encoder (Modules.py:312-385) - NOT part of the hot path (it produces one [B, 256]
conditioning vector per utterance, < 1 % of the FLOPs, "next" row 8f-3).  Kept as plain PyTorch-ROCm modules with the
reference's parameter names so that PE-mode checkpoints load and BASELINE config 5 runs; it feeds the HIP decoder /
encoder as a conditioning vector."""
import math

import torch


class _ConvBlock(torch.nn.Sequential):
    def __init__(self, cin, cout, k, stride):
        super().__init__()
        conv = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=(k - 1) // 2, bias=False)
        torch.nn.init.kaiming_uniform_(conv.weight, nonlinearity="relu")           # Modules.py:1010-1014
        self.add_module("Conv", conv)
        self.add_module("ReLU", torch.nn.ReLU(inplace=True))


class _Attention(torch.nn.Module):
    """RPR_Multihead_Attention without relative positions (Modules.py:349-355; RPR_MHA.py:69-128 with masks = None)."""

    def __init__(self, qc, kc, calc, out, heads):
        super().__init__()
        self.heads = heads
        self.layer_Dict = torch.nn.ModuleDict({
            "Query": torch.nn.Conv1d(qc, calc, 1), "Key": torch.nn.Conv1d(kc, calc, 1), "Value": torch.nn.Conv1d(kc, calc, 1),
            "Projection": torch.nn.Conv1d(calc, out, 1), "Dropout": torch.nn.Dropout(0.0)})
        for n in ("Query", "Key", "Value"):
            torch.nn.init.xavier_uniform_(self.layer_Dict[n].weight)

    def forward(self, queries, keys):
        B, _, Tq = queries.shape
        Tk = keys.shape[2]
        H = self.heads
        q = self.layer_Dict["Query"](queries)
        k = self.layer_Dict["Key"](keys)
        v = self.layer_Dict["Value"](keys)
        D = q.shape[1] // H
        q, k, v = (t.view(B, H, D, -1).transpose(2, 3) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(2, 3) / math.sqrt(D), dim=-1) @ v
        return self.layer_Dict["Projection"](a.transpose(2, 3).reshape(B, H * D, Tq))


class Prosody_Encoder(torch.nn.Module):
    def __init__(self, hp):
        super().__init__()
        pe = hp.Prosody_Encoder
        self.strides = list(pe.Reference_Encoder.Conv.Strides)
        self.layer_Dict = torch.nn.ModuleDict()
        cin, height = 1, hp.Sound.Mel_Dim
        for i, (k, c, s) in enumerate(zip(pe.Reference_Encoder.Conv.Kernel_Size, pe.Reference_Encoder.Conv.Channels, self.strides)):
            self.layer_Dict[f"Conv_{i}"] = _ConvBlock(cin, c, k, s)
            cin, height = c, math.ceil(height / s)
        self.layer_Dict["GRU"] = torch.nn.GRU(cin * height, pe.Reference_Encoder.GRU.Size, pe.Reference_Encoder.GRU.Stacks, batch_first=True)
        self.layer_Dict["Attention"] = _Attention(pe.Reference_Encoder.GRU.Size, pe.Style_Token.Size, pe.Size, pe.Size, pe.Style_Token.Attention_Head)
        self.gst_Tokens = torch.nn.Parameter(torch.randn(pe.Style_Token.Size, pe.Style_Token.Num_Tokens) * 0.5)
        self.n_conv = len(self.strides)

    def forward(self, x, lengths):
        x = x.unsqueeze(1)
        for i in range(self.n_conv):
            x = self.layer_Dict[f"Conv_{i}"](x)
        x = x.reshape(x.size(0), x.size(1) * x.size(2), x.size(3))
        x = self.layer_Dict["GRU"](x.transpose(2, 1))[0]                                  # [B, T', G]
        idx = (torch.ceil(lengths / float(math.prod(self.strides))).long() - 1).clamp_min(0)   # Modules.py:373
        x = x[torch.arange(x.size(0), device=x.device), idx]                               # [B, G]
        keys = torch.tanh(self.gst_Tokens).unsqueeze(0).expand(x.size(0), -1, -1)
        return self.layer_Dict["Attention"](x.unsqueeze(2), keys).squeeze(2)
