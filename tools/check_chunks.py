"""GPU: the decoder in utterance chunks (decoder.TUNE["dec_chunks"]) against the single-chain run: z, log-determinants and all gradients
(no dropout: a chunk's rows are numbered from 0, its dropout hash is its own).  usage: python tools/check_chunks.py [B] [chunks]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from glow_tts_amd import decoder, hparams                    # noqa: E402
from glow_tts_amd.hparams import Recursive_Parse             # noqa: E402
from glow_tts_amd.modules import GlowTTS, MLE_Loss           # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mode = sys.argv[3] if len(sys.argv) > 3 else "Vanilla"
d = hparams.load_yaml(hparams.DEFAULT_YAML)
d["HIP_Precision"] = "bf16"
d["Mode"] = mode
torch.manual_seed(0)
model = GlowTTS(Recursive_Parse(d)).cuda().eval()
g = torch.Generator().manual_seed(1)
tokens = torch.randint(0, 35, (B, 120), generator=g).cuda()
mels = (torch.randn(B, 80, 800, generator=g) * 1.5).clamp(-4, 4).cuda()
tl = torch.randint(60, 121, (B,), generator=g).cuda()
ml = torch.randint(400, 801, (B,), generator=g).cuda()
spk = torch.randint(0, 109, (B,), generator=g).cuda() if mode == "SE" else None
with torch.no_grad():
    for f in model.layer_Dict["Decoder"].layer_Dict["Flows"]:
        end = f.layers[2].layer_Dict["End"]
        end.weight.copy_(torch.randn(end.weight.shape, generator=g).cuda() * 0.02)
    model(tokens, tl, mels, ml, spk, None, None)
mle = MLE_Loss(model.hp)


def run(chunks, fb, skip):
    decoder.TUNE.update(dec_chunks=chunks, fused_wn_bwd=fb, fused_wn_fwd_skip=skip)
    model.zero_grad(set_to_none=True)
    z, mm, ms, ld, dur, durt, attn, _ = model(tokens, tl, mels, ml, spk, None, None)
    loss = mle(z=z, mean=mm, std=ms, log_dets=ld, lengths=ml) + torch.nn.functional.mse_loss(dur, durt)
    loss.backward()
    torch.cuda.synchronize()
    return z.detach().clone(), ld.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


za, la, ga = run(1, 12, 0)
zb, lb, gb = run(n, 12, 0)
print("z equal", torch.equal(za, zb), "logdet equal", torch.equal(la, lb))
worst, ne = 0.0, 0
for k in ga:
    if not torch.equal(ga[k], gb[k]):
        ne += 1
        rel = ((ga[k] - gb[k]).abs().max() / ga[k].abs().max().clamp_min(1e-30)).item()
        if rel > worst:
            worst, wk = rel, k
print(f"{len(ga)} gradients, {ne} not bit-equal, worst rel {worst:.3e}" + (f" ({wk})" if ne else ""))
assert torch.equal(za, zb) and torch.equal(la, lb) and worst < 1e-5
print("CHUNKS OK")
