"""Checker for a FREE-RUNNING OA-DM sampling run against the CPU oracle (TEST ORACLE infrastructure - used by tests/
and, after its timed region, by bench.py; never by evodiff_amd).

A free-running trajectory cannot be compared token for token with an independently sampled one (one boundary-ambiguous
draw and the paths part), but every single draw of it can be checked: the state before step s of generate_oaardm
(evodiff/generate.py:193-211) is the final sample with the positions order[s:] masked again, so the oracle's forward on
that state gives the distribution the draw at order[s] was made from, and the pre-drawn uniform gives the token the
reference would have drawn - which must be the sampled one wherever the uniform is not within ``margin_tol`` (probability
mass) of a category boundary.
"""
from __future__ import annotations

import numpy as np
import torch

from . import bytenet_ref as R


def spread_steps(n_steps: int, count: int):
    """``count`` step indices over early, middle and late parts of a loop of ``n_steps`` (always the first two - the
    cache-filling step and the first incremental one - and the last two)."""
    fixed = {0, 1, n_steps - 2, n_steps - 1}
    more = np.linspace(2, n_steps - 3, max(count - len(fixed), 0)).round().astype(int).tolist()
    return sorted(s for s in fixed.union(more) if 0 <= s < n_steps)


@torch.no_grad()
def oadm_spot_check(oracle, final, order, uniforms, rows, steps, margin_tol, n_draw: int = R.OADM_DRAW,
                    mask_id: int = R.MASK_ID, start=None):
    """``final`` (B, L) int64 on the host: the tokens a sampler produced from ``start`` (default: all mask) with
    ``order`` (n_steps,) and ``uniforms`` (n_steps, B).  For every (row, step) of ``rows`` x ``steps`` the oracle is
    teacher-forced along that trajectory.  Returns {"draws": compared, "agree": equal, "near_boundary": skipped}."""
    final = torch.as_tensor(final).cpu()
    order = np.asarray(order)
    uniforms = torch.as_tensor(uniforms).cpu()
    rows = list(rows)
    draws = agree = near = 0
    zeros = torch.zeros(len(rows), dtype=torch.long)
    for s in steps:
        state = final[rows].clone()
        later = torch.as_tensor(order[s:].astype(np.int64))
        state[:, later] = mask_id if start is None else torch.as_tensor(start)[rows][:, later]
        pos = int(order[s])
        logits = oracle(state, zeros)[:, pos, :n_draw]
        idx, margin = R.inverse_cdf(torch.softmax(logits, dim=-1), uniforms[s, rows])
        safe = margin > margin_tol
        near += int((~safe).sum())
        draws += int(safe.sum())
        agree += int((final[rows, pos][safe] == idx[safe]).sum())
    return {"draws": draws, "agree": agree, "near_boundary": near}
