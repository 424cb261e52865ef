"""On-device sampling (mi_forward_tokens / mi_op_sample) against its CPU restatement
(oracle/sampling.py) and against the CPU-sampling path.

Bar: greedy ids are bit-exact (== argmax of the logits mi_forward returns, lowest index on ties);
a sampled id equals the oracle's pick for the same (logits, params, seed, row) unless the uniform
draw lands within 1e-4 (relative) of a cumulative-mass boundary, where one fp32 rounding of an exp
may legitimately move it to the neighbouring candidate; the empirical distribution over many seeds
matches the nucleus distribution.  Parity with the reference: unpinned (no reference test holds
sampled ids; NxDI is absent) -- see oracle/sampling.py.
"""
import numpy as np
import pytest
import torch

from oracle import sampling as osamp
from oracle.synth import make_prompts, make_weights, zoo_config
from tests.test_model_gpu import load_golden, native_model, scenario

pytestmark = pytest.mark.gpu


def _op_sample_one(logits, params, seed, ws):
    from vllm_neuron_amd import _native
    L = _native.load_library()
    B, V = logits.shape
    ld = logits.cuda()
    pd = params.cuda() if params is not None else None
    out = torch.empty(B, dtype=torch.int32, device="cuda")
    if ws:   # the engine's form: with the scratch buffer the vocabulary is pre-selected by 32 work-groups per row
        nb = L.mi_op_sample_scratch_bytes(B)
        scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
        _native.check(L.mi_op_sample_ws(ld.data_ptr(), B, V, pd.data_ptr() if pd is not None else None, seed, out.data_ptr(),
                                        scratch.data_ptr(), nb, None))
    else:
        _native.check(L.mi_op_sample(ld.data_ptr(), B, V, pd.data_ptr() if pd is not None else None, seed, out.data_ptr(), None))
    torch.cuda.synchronize()
    return out.cpu().tolist()


def _op_sample(logits, params, seed):
    """Both forms of the sampler (one work-group per row / the engine's, spread over the chip): they must give the same ids."""
    a = _op_sample_one(logits, params, seed, False)
    b = _op_sample_one(logits, params, seed, True)
    assert a == b, (a, b)
    return a


def _agrees(logits_row, tk, tp, tt, seed, row, got):
    want = osamp.sample_row(logits_row, tk, tp, tt, seed, row)
    if got == want:
        return True
    idx, w = osamp.kept_distribution(logits_row, tk, tp, tt)
    idx = idx.tolist()
    if got not in idx:
        return False
    cum = np.cumsum(w.astype(np.float64))
    target = float(osamp.uniform(seed, row)) * cum[-1]
    i, j = sorted((idx.index(got), idx.index(want)))
    return j == i + 1 and abs(cum[i] - target) <= 1e-4 * cum[-1]


@pytest.mark.parametrize("V", [512, 4099, 128256])
def test_greedy_is_first_argmax(V):
    g = torch.Generator().manual_seed(V)
    logits = torch.randn(4, V, generator=g) * 3
    logits[1, 7] = logits[1].max() + 1          # unique maximum
    logits[2, [V - 1, 11, V // 2]] = 50.0       # ties: the lowest index wins
    logits[3] = -1e30
    logits[3, V - 3] = -5.0
    got = _op_sample(logits, None, 0)
    assert got == logits.argmax(dim=1).tolist()
    assert got[2] == 11
    params = torch.tensor([[1.0, 0.9, 0.7]] * 4)                      # top_k = 1 is greedy whatever the rest says
    assert _op_sample(logits, params, 123) == got


@pytest.mark.parametrize("V", [512, 128256])
def test_sampled_ids_follow_the_stated_rule(V):
    g = torch.Generator().manual_seed(7 + V)
    logits = torch.randn(4, V, generator=g) * 2.5
    logits[0, :5] = 6.0                                               # ties inside the top-k
    cases = [(50, 1.0, 1.0), (256, 0.9, 0.8), (2, 1.0, 1.5), (40, 0.3, 1.0), (1000, 0.95, 0.7)]
    bad = []
    for seed in range(40):
        for tk, tp, tt in cases:
            params = torch.tensor([[float(tk), tp, tt]] * 4)
            got = _op_sample(logits, params, seed)
            for row in range(4):
                if not _agrees(logits[row].numpy(), tk, tp, tt, seed, row, got[row]):
                    bad.append((seed, tk, tp, tt, row, got[row]))
    assert not bad, bad[:5]
    # same inputs, same ids
    params = torch.tensor([[50.0, 0.9, 1.0]] * 4)
    assert _op_sample(logits, params, 99) == _op_sample(logits, params, 99)


@pytest.mark.parametrize("V", [4099, 128256])
def test_mass_ties_at_the_threshold_never_evict_larger_logits(V):
    """More exact ties at the top-k threshold than the candidate buffer holds (1024): the sampler
    must keep every strictly larger logit and, of the ties, the lowest vocabulary indices --
    whatever the order the threads meet them in (the arrival-order collection could
    drop larger logits).  With top_p = 1 every kept candidate can be drawn: over many seeds the
    drawn ids must stay inside the oracle's kept set and cover the large logits."""
    g = torch.Generator().manual_seed(V)
    logits = torch.full((2, V), -4.0)
    tie_idx = torch.randperm(V, generator=g)[:3000]                  # 3000 exact ties at 1.0
    logits[:, tie_idx] = 1.0
    rest = torch.tensor(sorted(set(range(V)) - set(tie_idx.tolist())))
    big = rest[torch.randperm(len(rest), generator=g)[:10]]
    logits[0, big] = torch.linspace(2.0, 3.0, len(big))               # strictly above the ties
    tk = 50
    keep, _ = osamp.kept_distribution(logits[0].numpy(), tk, 1.0, 1.0)
    keep = set(keep.tolist())
    lowest_ties = sorted(tie_idx.tolist())[:tk - len(big)]
    assert keep == set(big.tolist()) | set(lowest_ties)
    params = torch.tensor([[float(tk), 1.0, 1.0]] * 2)
    seen = set()
    for seed in range(300):
        got = _op_sample(logits, params, seed)
        assert got[0] in keep, (seed, got[0])
        assert _agrees(logits[0].numpy(), tk, 1.0, 1.0, seed, 0, got[0])
        assert got == _op_sample(logits, params, seed)               # deterministic
        seen.add(got[0])
    assert seen & set(big.tolist())                                   # the large logits are drawn (they carry ~45 % of the mass)


@pytest.mark.parametrize("V", [8192, 20011, 128256, 262144])
def test_spread_sampler_edge_rows(V):
    """The chip-wide form on rows that stress the slice pre-selection: ties that straddle slice boundaries, a row whose
    finite logits are fewer than top_k (the rest -inf), the whole mass in the last slice, per-row top_k from 1 to 256."""
    g = torch.Generator().manual_seed(11 + V)
    per = -(-V // 32)
    logits = torch.randn(6, V, generator=g) * 2.0
    logits[0, per - 3:per + 3] = 9.0                       # six equal maxima across the first slice boundary
    logits[0, 5 * per - 1] = 9.0
    logits[1] = float("-inf")
    logits[1, torch.randperm(V, generator=g)[:7]] = torch.randn(7, generator=g)   # 7 finite words, top_k 40
    logits[2] = -30.0
    logits[2, V - 300:] = torch.randn(300, generator=g)    # everything that matters in the last slice
    logits[3, ::per] = 4.0                                 # one equal candidate at the head of every slice
    params = torch.tensor([[4.0, 1.0, 1.0], [40.0, 1.0, 1.0], [256.0, 0.95, 0.9], [20.0, 1.0, 1.2], [1.0, 1.0, 1.0], [100.0, 0.8, 0.7]])
    bad = []
    for seed in range(25):
        got = _op_sample(logits, params, seed)
        for row in range(6):
            tk, tp, tt = params[row].tolist()
            if not _agrees(logits[row].numpy(), int(tk), tp, tt, seed, row, got[row]):
                bad.append((seed, row, got[row]))
    assert not bad, bad[:5]
    assert _op_sample(logits, None, 0) == logits.argmax(dim=1).tolist()


def test_empirical_distribution_matches_the_nucleus():
    V, tk, tp, tt = 512, 8, 0.9, 1.3
    logits = torch.randn(1, V, generator=torch.Generator().manual_seed(3)) * 2
    idx, w = osamp.kept_distribution(logits[0].numpy(), tk, tp, tt)
    p = (w / w.sum()).astype(np.float64)
    params = torch.tensor([[float(tk), tp, tt]])
    n = 4000
    counts = dict.fromkeys(idx.tolist(), 0)
    for seed in range(n):
        t = _op_sample(logits, params, 1000 + seed)[0]
        assert t in counts, t                    # never outside the nucleus
        counts[t] += 1
    freq = np.array([counts[i] / n for i in idx.tolist()])
    assert np.abs(freq - p).max() < 4 * np.sqrt(0.25 / n) + 1e-3, (freq, p)


@pytest.mark.parametrize("name", ["llama31_like", "qwen25_like"])
def test_forward_tokens_equals_cpu_sampling_of_forward_logits(name):
    cfg = zoo_config(name)
    w = make_weights(cfg, seed=1)
    gen, _, _ = load_golden(name)
    prompts = make_prompts(cfg.vocab_size, 0)
    m = native_model(cfg, w, "f8e4m3", "per_channel_symmetric")
    for step, (kind, inp, rows) in enumerate(scenario(prompts, gen)):
        logits = m.forward(**inp)
        assert m.forward_tokens(**inp).tolist() == logits.argmax(dim=1).tolist(), (kind, step)
        B = logits.shape[0]
        params = torch.tensor([[20.0, 0.9, 0.8], [1.0, 1.0, 1.0], [256.0, 1.0, 1.0], [5.0, 0.5, 2.0]])[:B]
        got = m.forward_tokens(**inp, sampling_params=params, seed=step).tolist()
        for row in range(B):
            tk, tp, tt = params[row].tolist()
            # context encoding samples one sequence per launch: its row index is its place in the batch
            assert _agrees(logits[row].numpy(), tk, tp, tt, step, row, got[row]), (kind, step, row)
    with pytest.raises(ValueError):
        m.forward_tokens(**inp, sampling_params=torch.tensor([[0.0, 1.0, 1.0]] * B))     # top_k < 1
    m.close()
