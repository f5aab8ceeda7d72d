"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's optimizer side of the step - Rectified Adam (`Radam.py:25-90`), the
Modified Noam learning-rate schedule (`Noam_Scheduler.py:17-29`) and global-norm gradient clipping (`Train.py:228-231`,
torch.nn.utils.clip_grad_norm_).  Only tests / smoke / the bench's cpu_baseline may import this; the product path
(glow_tts_amd/optim.py + csrc/param_ops.hip) never does.  Pinned against the reference's own Radam.py / Noam_Scheduler.py by
tests/golden/make_optim_golden.py (fixture tests/golden/radam_case.npz)."""
import math

import numpy as np


def radam_scalars(step, beta1, beta2):
    """(N_sma, step_size) of Radam.py:63-79 for the 1-based step count."""
    beta2_t = beta2 ** step
    n_sma_max = 2.0 / (1.0 - beta2) - 1.0
    n_sma = n_sma_max - 2.0 * step * beta2_t / (1.0 - beta2_t)
    if n_sma >= 5:                                                                   # Radam.py:73-76
        step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma * n_sma_max / (n_sma_max - 2)) \
            / (1 - beta1 ** step)
    else:                                                                            # Radam.py:77-78
        step_size = 1.0 / (1 - beta1 ** step)
    return n_sma, step_size


def radam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """One update of one tensor (numpy float32 arrays, returned new): Radam.py:45-90.  `step` is the count AFTER the increment of :62."""
    f = np.float32
    g = g.astype(f)
    v = v * f(beta2) + f(1 - beta2) * g * g                                          # :59
    m = m * f(beta1) + f(1 - beta1) * g                                              # :60
    n_sma, step_size = radam_scalars(step, beta1, beta2)
    p = p.astype(f)
    if weight_decay != 0:
        p = p + f(-weight_decay * lr) * p                                            # :81-82
    if n_sma >= 5:
        p = p + f(-step_size * lr) * (m / (np.sqrt(v) + f(eps)))                     # :85-87
    else:
        p = p + f(-step_size * lr) * m                                               # :88-89
    return p.astype(f), m.astype(f), v.astype(f)


def modified_noam_lr(base_lr, base, last_epoch):
    """Noam_Scheduler.py:26-29."""
    e = max(1, last_epoch)
    return base_lr * base ** 0.5 * (e + base) ** (-0.5)


def noam_lr(base_lr, warmup_steps, last_epoch):
    """Noam_Scheduler.py:10-14."""
    e = max(1, last_epoch)
    return base_lr * warmup_steps ** 0.5 * min(e ** (-0.5), e * warmup_steps ** -1.5)


def clip_coef(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (Train.py:228-231): (total L2 norm, min(1, max_norm / (norm + 1e-6)))."""
    total = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads))
    return total, min(1.0, max_norm / (total + 1e-6))
