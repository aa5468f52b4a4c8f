"""-m gpu: the real TrainEngine on TWO ranks (torch.distributed.run; both on cuda:0 over gloo on the 1-GPU test box).

One step, rank r on its own local batch, rank 1 deliberately started from different weights.  Checked:
  * replicas are identical after the step (parameters and EMA bit-equal across ranks) -- i.e. the engine broadcast rank 0's
    weights at construction and every rank applied the same reduced gradient;
  * the post-step parameters equal the CPU oracle's step on the MEAN of the two local-batch gradients (each D: mean gradient
    -> Adam; then the generator loss through the updated Ds, mean gradient -> Adam -> EMA), SURVEY.md section 8(e):
    judged on parameter deltas (Adam's first step is lr * sign(g); see helpers.AdamDeltaCheck for the tolerance rationale).
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import det_state, load_pkg
from standin import StandInEncoder

load_pkg()
from mogan_amd.attngan import synthetic  # noqa: E402
from oracle import attngan_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_mean_gradient_step(steps=1):
    cfg = O.Cfg(gf_dim=4, df_dim=4, emb_dim=16, r_num=2, words_num=5)
    from helpers import det_fill_state
    G = O.from_state_dict(det_state(O.g_net_spec(cfg), "G."))
    Ds = [O.from_state_dict(det_state(O.d_net_spec(i, cfg), "D%d." % i)) for i in range(3)]
    enc = StandInEncoder(16)
    det_fill_state(enc, "ENC.")
    for p in enc.parameters():
        p.requires_grad = False
    enc.eval()
    st = O.TrainState(G, Ds, cfg)
    init = {"G": {k: v.detach().clone() for k, v in G.items()},
            "D": [{k: v.detach().clone() for k, v in d.items()} for d in Ds]}
    for step in range(steps):
        batches = [synthetic.make_batch(4, words_num=5, nef=16, seed=100 + 10 * step + r) for r in range(2)]
        outs = [O.g_net(st.g, cfg, b["z"], b["sent_emb"], b["words_embs"], b["mask"], b["tmi"], b["label_one_hot"], b["eps"])
                for b in batches]
        for i, d in enumerate(st.ds):
            O.zero_grad(d)
            for b, o in zip(batches, outs):
                (O.discriminator_loss(i, d, b["imgs"][i], o[0][i], b["sent_emb"], b, cfg) / 2.0).backward()
            O.adam_step(d, st.opt_ds[i], cfg.lr_d)
        O.zero_grad(st.g)
        for b, o in zip(batches, outs):
            err_g, _ = O.generator_loss(st.ds, enc, o[0], b, cfg)
            ((err_g + O.kl_loss(o[2], o[3])) / 2.0).backward()
        O.adam_step(st.g, st.opt_g, cfg.lr_g)
    return init, st


@pytest.mark.parametrize("branch_graphs", ["0", "1"], ids=["eager", "branch-graphs"])
@pytest.mark.parametrize("chunk_mb", ["0", "0.05"], ids=["whole-bucket", "chunked-overlap"])
def test_two_rank_engine_step_equals_oracle_mean_gradient_step(tmp_path, chunk_mb, branch_graphs):
    """chunk_mb 0.05: every bucket of the reduced-width networks is cut into several chunks whose all-reduces start during
    the backward pass (trainer.ChunkedReducer); two steps, so that the second one runs with the learned schedule.
    branch-graphs (the default under data parallelism): the discriminator branches are replayed hipGraphs cut between the
    backward and Adam, their buckets are reduced between the two replays (ChunkedReducer.reduce_now / one all-reduce)."""
    env = dict(os.environ, MOGAN_DIST_BACKEND="gloo", MOGAN_STREAMS=os.environ.get("MOGAN_STREAMS", "1"),
               MOGAN_DP_CHUNK_MB=chunk_mb, DP_STEPS="2", MOGAN_BRANCH_GRAPHS_DP=branch_graphs)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(29900 + os.getpid() % 90), os.path.join(ROOT, "tests", "dp_worker.py"),
           str(tmp_path)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r0 = torch.load(str(tmp_path / "rank0.pt"), weights_only=False)
    r1 = torch.load(str(tmp_path / "rank1.pt"), weights_only=False)
    # (1) identical replicas (BatchNorm running buffers are per replica by design)
    for k, v in r0["G"].items():
        if "running" not in k and "num_batches" not in k:
            assert torch.equal(v, r1["G"][k]), "G %s differs between the ranks" % k
    for i in range(3):
        for k, v in r0["D"][i].items():
            if "running" not in k and "num_batches" not in k:
                assert torch.equal(v, r1["D"][i][k]), "D%d %s differs between the ranks" % (i, k)
    assert torch.equal(r0["ema"], r1["ema"])
    assert r0["logs"]["errD0"] != r1["logs"]["errD0"]            # ... although they saw different batches
    if chunk_mb != "0":
        # every bucket was cut into several chunks and, in the second step, all but (at most) the last one left while the
        # backward pass was still being queued
        assert len(r0["reducers"]) == 4 and all(n >= 2 for n, _ in r0["reducers"]), r0["reducers"]
        hooked = r0["reducers"] if branch_graphs == "0" else r0["reducers"][:1]      # (the generator's comes first)
        assert all(early >= n - 1 for n, early in hooked), r0["reducers"]
        if branch_graphs != "0":                      # replayed backward: nothing can leave before the graph has been queued
            assert all(early == 0 for _, early in r0["reducers"][1:]), r0["reducers"]
        print("chunks (count, started during the backward):", r0["reducers"])
    else:
        assert r0["reducers"] == []
    # (2) = the oracle's step on the mean gradient
    init, st = _oracle_mean_gradient_step(steps=2)
    lr, nsteps = 2e-4, 2
    report, failures = [], []
    for name, got_sd, want_net, init_net, max_bad in [("G", r0["G"], st.g, init["G"], 0.02)] + \
            [("D%d" % i, r0["D"][i], st.ds[i], init["D"][i], 0.01) for i in range(3)]:
        n = bad = moved = 0
        for k, w in want_net.items():
            if not (torch.is_tensor(w) and w.requires_grad):
                continue
            d_got = (got_sd[k].double() - init_net[k].double()).flatten()
            d_want = (w.detach().double() - init_net[k].double()).flatten()
            n += d_want.numel()
            bad += int(((d_got - d_want).abs() > 0.5 * lr).sum())
            moved += int((d_want.abs() > 0.5 * lr).sum())
            if float(d_got.abs().max()) > nsteps * lr * 1.2:      # (Adam's second step can exceed lr slightly)
                failures.append("%s %s moved by more than lr" % (name, k))
        report.append("%s: %d parameters, %.2f%% of the deltas off by > lr/2 (allowed %.0f%%), %.0f%% moved by > lr/2"
                      % (name, n, 100.0 * bad / n, 100 * max_bad, 100.0 * moved / n))
        if moved <= 0.5 * n:
            failures.append("%s: the oracle moved only %d of %d elements" % (name, moved, n))
        if bad > max_bad * n:
            failures.append("%s: %d of %d parameter deltas differ from the mean-gradient step by > lr/2" % (name, bad, n))
    print("\n".join(report))
    assert not failures, "; ".join(failures) + " | " + " | ".join(report)


@pytest.mark.parametrize("branch_graphs", ["0", "1"], ids=["eager", "branch-graphs"])
def test_two_rank_full_width_shard(tmp_path, branch_graphs):
    """BASELINE config 4's literal shard on two ranks: the benchmark's full-width networks (coco_train.yml, the real Inception
    encoder, B = 4 per rank, both ranks on cuda:0 over gloo), ranks started from different weights, two steps.  The replicas
    must be identical afterwards (checksums of every flat parameter bucket and of the EMA shadow), finite, different from each
    other in what they saw; the generator's bucket and D_NET256's travel in chunks that start during their backward passes."""
    env = dict(os.environ, MOGAN_DIST_BACKEND="gloo", DP_FULL="1", MOGAN_BRANCH_GRAPHS_DP=branch_graphs)
    env.pop("MOGAN_DP_CHUNK_MB", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(29990 - os.getpid() % 90), os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r0 = torch.load(str(tmp_path / "rank0.pt"), weights_only=False)
    r1 = torch.load(str(tmp_path / "rank1.pt"), weights_only=False)
    for k, v in r0["sums"].items():
        assert v == r1["sums"][k], "%s differs between the ranks: %s vs %s" % (k, v, r1["sums"][k])
        if len(v) == 4:
            assert v[3] and v[2] > 0, "%s: not finite / Adam moments empty" % k
    assert r0["logs"]["errD2"] != r1["logs"]["errD2"]
    assert all(abs(x) < 1e6 for x in r0["logs"].values())
    assert set(r0["reducers"]) == {"G", "D2"}, r0["reducers"]
    for name, (n, early) in r0["reducers"].items():
        assert n >= 2 and (early >= n - 1 if (name == "G" or branch_graphs == "0") else early == 0), (name, n, early)
