"""CPU-only tests of the host logic: reference-compatible state_dict tables, the drop-in
module paths, image sharding / detection gather (gloo, world_size 2), product/oracle
separation."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tag", ["base_nano", "base_tiny", "base_s", "gl_nano", "gl_tiny", "gl_s"])
def test_state_dict_tables_equal_the_reference(shapes, tag):
    """keys, shapes AND order of the reference's state_dict (tests/golden/shapes.json was
    dumped from the imported reference modules)"""
    from glsdet_amd.arch import state_dict_shapes
    kind, phi = tag.split("_")
    mine = state_dict_shapes(kind, phi, 10)
    assert list(mine.keys()) == list(shapes[tag].keys())
    for k, s in shapes[tag].items():
        assert tuple(s) == tuple(mine[k]), k


def test_arch_rejects_unknown():
    from glsdet_amd.arch import state_dict_shapes
    with pytest.raises(ValueError):
        state_dict_shapes("nope", "s", 10)
    with pytest.raises(KeyError):
        state_dict_shapes("gl", "xxl", 10)


def test_drop_in_module_paths_like_the_reference_harness(shapes, monkeypatch):
    """drone/yolo.py:99-105: importlib on a config path string, YoloBody(nc, phi),
    load_state_dict, eval, DataParallel"""
    monkeypatch.syspath_prepend(os.path.join(ROOT, "glsdet_amd", "drone"))
    for mod in list(sys.modules):
        if mod == "models" or mod.startswith("models."):
            monkeypatch.delitem(sys.modules, mod)
    m6 = importlib.import_module("models/new/yolox6.py"[:-3].replace("/", "."))
    assert len(m6.YoloBody(10, "tiny").state_dict()) == 540
    for path, tag in (("models/base/yolox.py", "base_tiny"),
                      ("models/block/non_local/yolo_patch_nonlocal_plus.py", "gl_tiny")):
        m = importlib.import_module(path[:-3].replace("/", "."))
        net = m.YoloBody(10, "tiny")
        assert list(net.state_dict().keys()) == list(shapes[tag].keys())
        from glsdet_amd.synth import synth_state_dict
        sd = synth_state_dict(shapes[tag], 0)
        net.load_state_dict(sd)
        back = net.state_dict()
        assert all(torch.equal(back[k], sd[k]) for k in sd)
        net.load_state_dict({"module." + k: v for k, v in sd.items()})      # DataParallel checkpoint
        with pytest.raises(RuntimeError):
            net.load_state_dict({"bogus": torch.zeros(1)})
        net = net.eval()
        assert isinstance(torch.nn.DataParallel(net).module, type(net))
        with pytest.raises(NotImplementedError):
            net.train()(torch.zeros(1, 3, 64, 64))
    ub = importlib.import_module("models.core.utils_bbox")
    assert callable(ub.decode_outputs) and callable(ub.non_max_suppression)


def test_yolo_correct_boxes_twin_matches_reference(golden):
    from glsdet_amd.drone.models.core.utils_bbox import yolo_correct_boxes
    for lb in (0, 1):
        got = yolo_correct_boxes(golden["correct_boxes/xy"].copy(), golden["correct_boxes/wh"].copy(),
                                 [640, 640], np.array([540, 1024]), bool(lb))
        np.testing.assert_allclose(got, golden["correct_boxes/letterbox%d" % lb], rtol=1e-5, atol=1e-3)


def test_product_never_imports_the_oracle():
    """the shipped path must not route through the checker"""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "glsdet_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                if "import oracle" in src or "from oracle" in src:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_shard_indices_partition():
    from glsdet_amd.dist import shard_indices
    for world in (1, 2, 3, 8):
        parts = [shard_indices(19, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(19))
        assert all(p == list(range(r, 19, world)) for r, p in enumerate(parts))


_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from glsdet_amd.dist import gather_detections, unpack_in_dataset_order
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=world)
n, K = 3, 5
rng = np.random.default_rng(100 + rank)
count = torch.tensor([2, 0, 5] if rank == 0 else [1, 4, 3], dtype=torch.int32)
dets = torch.zeros(n, K, 7)
for i in range(n):
    dets[i, : count[i]] = torch.from_numpy(rng.uniform(0, 1, (int(count[i]), 7)).astype(np.float32)) + 10 * rank + i
g = gather_detections(dets, torch.cat([count, count]))
assert tuple(g.shape) == (world, n, K + 1, 7)
res = unpack_in_dataset_order(g, num_images=5)       # 6 slots, the last one is sampler padding
assert len(res) == 5
want_counts = [2, 1, 0, 4, 5]                         # image i -> rank i %% 2, slot i // 2
assert [len(r) for r in res] == want_counts, [len(r) for r in res]
for i, r in enumerate(res):
    if len(r):
        assert int(r[0, 0]) // 10 == i %% 2 and int(r[0, 0]) %% 10 == i // 2
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_gather_detections_gloo_world2(tmp_path):
    """N>1 path on CPU: one all_gather of fixed-capacity detections, dataset order restored"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(_WORKER % {"root": ROOT, "port": port})
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


_AGREE_WORKER = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from glsdet_amd.dist import ranks_agree
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=world)
# a clock-driven loop with collectives inside (bench.py's settle phase): rank 1's clock runs out three rounds before
# rank 0's -- both must run the same number of rounds, or the gathers below pair with the barrier after the loop
budget = 6 if rank == 0 else 3
rounds = 0
buf = torch.empty(world, dtype=torch.int32)
while ranks_agree(rounds < budget, "any"):
    dist.all_gather_into_tensor(buf, torch.tensor([rounds], dtype=torch.int32))
    assert buf.tolist() == [rounds, rounds]
    rounds += 1
dist.barrier()
assert rounds == 6
# a failure on one rank ends the phase on both
assert ranks_agree(True, "all") is True
assert ranks_agree(rank == 0, "all") is False
assert ranks_agree(rank == 1, "any") is True
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_ranks_agree_gloo_world2(tmp_path):
    """the rule behind bench.py's N>1 control flow: no rank decides alone whether to enter a collective"""
    import socket
    from glsdet_amd.dist import ranks_agree
    assert ranks_agree(True) is True and ranks_agree(False, "any") is False         # no process group: the local answer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "a.py"
    script.write_text(_AGREE_WORKER % {"root": ROOT, "port": port})
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
