"""CPU-side checks of the boundary and host logic (no kernels are launched): the C-ABI library
loads and exports exactly the symbols include/mogan_hip.h declares, the ctypes table mirrors the
header, the module tree reproduces the reference's state_dict keys, cfg/yml handling, the synthetic
batch contract, and the product refuses to run without a GPU (no fallback)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from helpers import ROOT, load_pkg
from oracle import attngan_oracle as O

load_pkg()
from mogan_amd.attngan import synthetic  # noqa: E402
from mogan_amd.attngan.miscc import config as C  # noqa: E402
from mogan_amd.hip import lib, ops  # noqa: E402

HEADER = os.path.join(ROOT, "include", "mogan_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mogan_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    names = _declared()
    assert len(names) >= 40
    so = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(so, n), "libmogan_hip.so does not export %s" % n
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH]).decode()
    exported = sorted(set(re.findall(r" T (mogan_[a-z0-9_]+)", out)))
    assert exported == names, set(exported) ^ set(names)
    assert sorted(lib.SIGNATURES) == names                 # the ctypes table mirrors the header
    assert lib.load().mogan_abi_version() == 1


def test_ctypes_arity_matches_header():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, args in lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, src, flags=re.S)
        assert m, name
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(args), (name, len(params), len(args))


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """the argument structs of the grouped entry points (MoganConvFwdArgs, MoganConvDgradArgs, MoganPkArgs, MoganTailArgs): size and the
    offset of every field as gcc lays out include/mogan_hip.h against the ctypes mirrors in hip/lib.py"""
    pairs = (("MoganConvFwdArgs", lib.ConvFwdArgs), ("MoganConvDgradArgs", lib.ConvDgradArgs), ("MoganPkArgs", lib.PkArgs),
             ("MoganTailArgs", lib.TailArgs))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % HEADER, "int main(void) {"]
    for cname, cls in pairs:
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f[0], cname, f[0]))
    lines.append("return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in pairs:
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for f in cls._fields_:
            assert int(got["%s.%s" % (cname, f[0])]) == getattr(cls, f[0]).offset, (cname, f[0])


def test_no_cpu_fallback():
    with pytest.raises(lib.MoganHipError):
        ops.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3), None, 1, 1)
    with pytest.raises(lib.MoganHipError):
        ops.stn(torch.zeros(1, 1, 4, 4), torch.zeros(1, 2, 3), (1, 1, 4, 4))


def test_state_dict_keys_match_reference_layout():
    """Key names/shapes/order = what the reference's modules produce (oracle specs were checked against
    the reference golden run; spot-check of the names quoted in SURVEY.md §5)."""
    from mogan_amd.attngan import model
    C.set_coco_train_defaults()
    ocfg = O.Cfg()
    G = model.G_NET()
    assert [(k, tuple(v.shape)) for k, v in G.state_dict().items()] == \
        [(k, tuple(s)) for k, s in O.g_net_spec(ocfg).items()]
    for i, cls in enumerate((model.D_NET64, model.D_NET128, model.D_NET256)):
        D = cls()
        assert [(k, tuple(v.shape)) for k, v in D.state_dict().items()] == \
            [(k, tuple(s)) for k, s in O.d_net_spec(i, ocfg).items()]
    sd = G.state_dict()
    assert tuple(sd["h_net1.upsample1.1.weight"].shape) == (768, 768, 3, 3)
    assert tuple(sd["h_net2.att.conv_context.weight"].shape) == (48, 256, 1, 1)
    D = model.D_NET256().state_dict()
    assert tuple(D["img_code_s64.0.weight"].shape) == (3072, 1536, 4, 4)
    assert tuple(D["COND_DNET.jointConv.0.weight"].shape) == (768, 1024, 3, 3)
    n = lambda m: sum(p.numel() for p in m.parameters())
    assert (n(G), n(model.D_NET64()), n(model.D_NET128()), n(model.D_NET256())) == \
        (17420812, 14742530, 42800258, 160774274)
    # weights_init dispatches on class names (miscc/utils.py:321-331)
    from mogan_amd.attngan.miscc.utils import weights_init
    G.apply(weights_init)
    w = G.h_net1.upsample3[1].weight.detach().reshape(192, -1)
    torch.testing.assert_close(w @ w.t(), torch.eye(192), atol=1e-4, rtol=0)
    enc = model.CNN_ENCODER(256)
    keys = list(enc.state_dict())
    assert "Mixed_6e.branch7x7dbl_5.conv.weight" in keys and "Mixed_7c.branch3x3dbl_3b.bn.running_var" in keys
    assert "emb_features.weight" in keys and "emb_cnn_code.bias" in keys


def test_cfg_from_file_and_errors(tmp_path):
    C.cfg_from_file(os.path.join(ROOT, "multiple-objects-gan_amd", "attngan", "cfg", "coco_train.yml"))
    assert C.cfg.GAN.GF_DIM == 48 and C.cfg.GAN.DF_DIM == 96 and C.cfg.TEXT.WORDS_NUM == 12
    assert C.cfg.TRAIN.SMOOTH.LAMBDA == 50.0 and C.cfg.TRAIN.BATCH_SIZE == 14 and C.cfg.GPU_ID == '0,1,2'
    bad = tmp_path / "bad.yml"
    bad.write_text("NOT_A_KEY: 1\n")
    with pytest.raises(KeyError):
        C.cfg_from_file(str(bad))
    bad.write_text("GAN: {GF_DIM: 'x'}\n")
    with pytest.raises(ValueError):
        C.cfg_from_file(str(bad))


def test_synthetic_batch_contract():
    b = synthetic.make_batch(16, words_num=12, nef=256, seed=3, text="tokens")
    assert [tuple(t.shape) for t in b["imgs"]] == [(16, 3, 64, 64), (16, 3, 128, 128), (16, 3, 256, 256)]
    assert all(t.dtype == torch.float32 and float(t.abs().max()) <= 1 for t in b["imgs"])
    lens = b["cap_lens"]
    assert lens.dtype == torch.int64 and lens[0] == 12 and bool((lens[:-1] >= lens[1:]).all()) and int(lens.min()) >= 5
    cap = b["captions"]
    for i in range(16):
        assert bool((cap[i, :lens[i]] > 0).all()) and bool((cap[i, lens[i]:] == 0).all())
    assert bool((b["mask"] == (cap == 0)).all())
    assert tuple(b["tm"].shape) == (16, 3, 2, 3) and tuple(b["label_one_hot"].shape) == (16, 3, 81)
    assert bool((b["label_one_hot"].sum(-1) == 1).all())
    bbox = b["bbox"]
    present = bbox[..., 0] >= 0
    assert bool(((bbox[..., 0] + bbox[..., 2])[present] <= 0.9991).all())          # datasets.py:115-121
    absent = ~present
    assert bool((b["label_one_hot"][absent][:, 80] == 1).all())                      # -1 -> class 80
    np.testing.assert_array_equal(b["tmi"][absent][0].numpy(), np.array([[-1, 0, -4], [0, -1, -4]], np.float32))
    np.testing.assert_array_equal(b["class_ids"], np.arange(16))


def test_tuned_gemm_table_is_well_formed():
    """hip/tuned_gemm_gfx950.csv (tools/tune_all.sh): one `mode,M,N,K,nz,cfg,split,...` line per tuned GEMM; the loader
    hands the first seven integers to mogan_gemm_tune_set, which rejects anything outside these ranges."""
    import os
    path = os.path.join(ROOT, "multiple-objects-gan_amd", "hip", "tuned_gemm_gfx950.csv")
    rows = [l.split(",") for l in open(path) if l.strip() and not l.startswith("#")]
    assert len(rows) > 100
    seen = set()
    for r in rows:
        mode, M, N, K, nz, cfg, split = (int(x) for x in r[:7])
        assert 0 <= mode <= 3 and M > 0 and N > 0 and K > 0 and nz >= 1 and 0 <= cfg < 7 and 1 <= split <= 64
        assert (mode, M, N, K, nz) not in seen
        seen.add((mode, M, N, K, nz))


def test_workspace_is_separate_for_captured_launches(monkeypatch):
    """hip/lib.py:workspace -- launches recorded into a hipGraph must not share the split-K / Winograd scratch buffer with
    eager launches on a stream that happens to carry the capture stream's handle (torch records all graphs on one pool
    stream), nor with a graph of another capture session.  Driven on CPU with the stream / capture queries mocked."""
    import types
    import torch
    from mogan_amd.hip import lib
    state = {"handle": 7, "capturing": False}
    made = []

    def fake_empty(n, dtype=None, device=None):
        t = torch.zeros(16, dtype=torch.uint8)
        made.append(t)
        return t

    monkeypatch.setattr(lib, "_raw_stream", lambda idx: state["handle"])
    monkeypatch.setattr(lib, "_capturing", lambda: state["capturing"])
    monkeypatch.setattr(lib, "_ws", {})
    monkeypatch.setattr(lib, "_cap_epoch", [0, False])
    monkeypatch.setattr(lib.torch, "empty", fake_empty)
    dev = types.SimpleNamespace(type="cuda", index=0)
    eager = lib.workspace(dev)[0]
    assert lib.workspace(dev)[0] == eager and len(made) == 1
    state["capturing"] = True
    cap1 = lib.workspace(dev)[0]
    assert cap1 != eager and lib.workspace(dev)[0] == cap1           # one buffer per capture session
    state["handle"] = 9                                              # a second stream inside the same capture
    cap1b = lib.workspace(dev)[0]
    assert cap1b not in (eager, cap1)
    state["capturing"], state["handle"] = False, 7
    assert lib.workspace(dev)[0] == eager                            # eager work keeps its own buffer
    state["capturing"] = True
    cap2 = lib.workspace(dev)[0]
    assert cap2 not in (eager, cap1, cap1b) and len(made) == 4       # a later capture session: a new buffer
    with pytest.raises(lib.MoganHipError):
        lib.workspace(types.SimpleNamespace(type="cpu", index=None))


def test_hw_queue_configuration_respects_the_user(monkeypatch):
    """hip/lib.py:configure_hw_queues -- GPU_MAX_HW_QUEUES is only defaulted (4 queues, no idle streams first: ONE default for a
    single process and for a member of a process group since round 5, the engine's streams being created in a fixed order before
    the group exists), never overridden; reserve_hw_queues(0) and the default reserve_hw_queues() do not touch the library."""
    from mogan_amd.hip import lib
    for k in ("GPU_MAX_HW_QUEUES", "WORLD_SIZE", "MOGAN_FORCE_DIST"):
        monkeypatch.delenv(k, raising=False)
    assert lib.hw_queue_defaults() == ("4", 0)
    lib.configure_hw_queues()
    assert os.environ["GPU_MAX_HW_QUEUES"] == "4"
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert lib.hw_queue_defaults() == ("4", 0)
    lib.configure_hw_queues()
    assert os.environ["GPU_MAX_HW_QUEUES"] == "4"
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("MOGAN_FORCE_DIST", "1")
    assert lib.hw_queue_defaults() == ("4", 0)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "5")
    lib.configure_hw_queues()
    assert os.environ["GPU_MAX_HW_QUEUES"] == "5"
    monkeypatch.setattr(lib, "load", lambda: (_ for _ in ()).throw(AssertionError("library touched")))
    monkeypatch.setattr(lib, "_reserved", [])
    lib.reserve_hw_queues(0)
    lib.reserve_hw_queues()
    assert lib._reserved == []


def test_flat_adam_state_dict_is_torch_adam_layout_and_roundtrips():
    """--resume must bring back the Adam moments and step counters; checkpoints carry torch.optim.Adam's own
    state_dict layout (what the reference saves as optimG / optimD, trainer.py:188-196), and both that layout -- e.g. from a
    real torch.optim.Adam -- and the round-1 flat layout load."""
    import torch.nn as nn
    from mogan_amd.attngan.trainer import FlatAdam
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(5, 4), nn.Tanh(), nn.Linear(4, 3))
    ref = torch.optim.Adam(net.parameters(), lr=2e-4, betas=(0.5, 0.999))
    for _ in range(3):
        ref.zero_grad()
        net(torch.randn(6, 5)).pow(2).sum().backward()
        ref.step()
    want = ref.state_dict()
    flat = FlatAdam(net, lr=1e-3)
    flat.load_state_dict(want)                                   # torch layout in
    assert float(flat.state[0]) == 3.0 and flat.lr == 2e-4
    for i, (p, o) in enumerate(zip(flat.params, flat.offsets)):
        torch.testing.assert_close(flat.m[o:o + p.numel()].view_as(p), want["state"][i]["exp_avg"])
        torch.testing.assert_close(flat.v[o:o + p.numel()].view_as(p), want["state"][i]["exp_avg_sq"])
    got = flat.state_dict()                                      # torch layout out
    assert set(got) == {"state", "param_groups"} and got["param_groups"][0]["betas"] == (0.5, 0.999)
    ref2 = torch.optim.Adam(net.parameters(), lr=2e-4, betas=(0.5, 0.999))
    ref2.load_state_dict(got)                                    # ... which a real torch optimizer accepts
    assert float(ref2.state_dict()["state"][0]["step"]) == 3.0
    flat2 = FlatAdam(net, lr=2e-4)
    flat2.load_state_dict({"step": 3.0, "exp_avg": flat.m.clone(), "exp_avg_sq": flat.v.clone(), "lr": 2e-4})   # round-1 layout
    assert torch.equal(flat2.m, flat.m) and float(flat2.state[0]) == 3.0


def test_bench_self_launch_builds_the_torchrun_command(monkeypatch):
    """`python bench.py --gpus N` (N > 1) without a launcher environment replaces itself by torch.distributed.run with N
    ranks on 127.0.0.1 and the same arguments; with RANK/WORLD_SIZE set (the driver's own launch) or N = 1 it does nothing."""
    import sys
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index('if __name__ == "__main__":\n    _self_launch()')]
    ns = {"__file__": os.path.join(ROOT, "bench.py")}
    exec(compile(head, "bench_head", "exec"), ns)
    got = []
    monkeypatch.setattr(ns["os"], "execv", lambda exe, argv: got.append((exe, argv)))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    ns["_self_launch"]()
    assert len(got) == 1
    argv = got[0][1]
    assert argv[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in argv
    assert argv[argv.index("--nproc-per-node") + 1] == "8" and argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"] and argv[-7].endswith("bench.py")
    del got[:]
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus=1"])
    ns["_self_launch"]()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    monkeypatch.setenv("RANK", "0")
    ns["_self_launch"]()
    assert got == []


def test_engine_stream_order_is_a_fixed_list_of_known_streams():
    """trainer.ENGINE_STREAM_ORDER (the order in which every entry point creates and first uses the engine's streams, which alone
    decides the stream -> hardware-queue layout): every token names a stream of the engine, the
    streams the step actually runs on are all in it exactly once, and the D_NET256 branch comes first."""
    from mogan_amd.attngan import trainer
    toks = trainer.ENGINE_STREAM_ORDER.split(",")
    known = {"s0", "s1", "s2", "s3", "w0", "w1", "w2", "wm", "gc", "cG", "cD", "x"}
    assert all(t in known for t in toks), toks
    for t in ("s0", "s1", "s2", "s3", "wm", "gc"):
        assert toks.count(t) == 1, t
    assert toks[0] == "s2"
