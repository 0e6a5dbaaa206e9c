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


def _reference_uninterleave(part_list, size):
    """collect_results_cpu / collect_results_gpu, /root/reference/yolox-ufp/mmdet/apis/test.py:150-155,186-190, restated:
    `for res in zip(*part_list): ordered_results.extend(list(res))`, then `ordered_results[:size]`."""
    ordered = []
    for res in zip(*part_list):
        ordered.extend(list(res))
    return ordered[:size]


@pytest.mark.parametrize("world,per_rank,size", [(8, 8, 64), (8, 8, 61), (3, 7, 19), (2, 3, 5), (4, 1, 3)])
def test_unpack_in_dataset_order_equals_the_reference_uninterleave(world, per_rank, size):
    """BASELINE config 4 (64 images over 8 GPUs) and counts the world size does not divide: the exchange record unpacked
    in data-set order equals the reference's zip / extend / [:size] on per-rank result lists (DistributedSampler order,
    the sampler's padding images dropped)."""
    import numpy as np
    import torch
    from glsdet_amd.dist import shard_indices, unpack_in_dataset_order
    cap = 6
    rng = np.random.default_rng(world * 100 + size)
    g = np.zeros((world, per_rank, cap + 1, 7), np.float32)
    part_list = []
    for r in range(world):
        part = []
        for slot in range(per_rank):
            k = int(rng.integers(0, cap + 1))
            rows = rng.uniform(0, 1, (k, 7)).astype(np.float32)
            rows[:, 0] = 1000 * r + slot                  # tag: where the rows came from
            g[r, slot, :k] = rows
            g[r, slot, cap, 0] = k
            part.append(rows)
        part_list.append(part)
    want = _reference_uninterleave(part_list, size)
    got = unpack_in_dataset_order(torch.from_numpy(g), num_images=size)
    assert len(got) == len(want) == size
    for i, (a, b) in enumerate(zip(got, want)):
        np.testing.assert_array_equal(a, b)
        if len(a):
            assert int(a[0, 0]) == 1000 * (i % world) + i // world
    # and the sampler side: image i is handled by rank i % world at slot i // world
    for r in range(world):
        assert shard_indices(size, r, world) == [i for i in range(size) if i % world == r]


_TUNE_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from glsdet_amd import engine
from glsdet_amd.dist import share_tuning
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert share_tuning(0) == 0                              # no process group yet: a no-op
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=world)
if rank == 0:                                            # what Engine(autotune=True) leaves behind on the tuning rank
    engine._SHARED_TUNED["f16"] = {(8, 100, 168, 128, 1, True, 0): 0x20d, ("chain", 8, 50, 84, 256): -1, ("multi", 4, 1, 1, 3, 3, 8): 0x108}
    engine._SHARED_TUNED["f32"] = {(1, 25, 42, 512): 9}
else:                                                    # a rank that tuned something on its own keeps it unless rank 0 says otherwise
    engine._SHARED_TUNED["f16"] = {(8, 100, 168, 128, 1, True, 0): 13, ("own", 1): 5}
n = share_tuning(0)
t = engine._SHARED_TUNED
assert t["f16"][(8, 100, 168, 128, 1, True, 0)] == 0x20d and t["f16"][("chain", 8, 50, 84, 256)] == -1 and t["f32"][(1, 25, 42, 512)] == 9
assert t["f16"][("multi", 4, 1, 1, 3, 3, 8)] == 0x108
assert (("own", 1) in t["f16"]) == (rank == 1) and n == (4 if rank == 0 else 5)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_share_tuning_gloo_world2(tmp_path):
    """every rank runs the kernel variants rank 0 measured (bench.py N > 1): one object broadcast of the tuning table"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "t.py"
    script.write_text(_TUNE_WORKER % {"root": ROOT, "port": port})
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


@pytest.mark.parametrize("spec,want_rc", [("ok", 0), ("die:1:3", 3), ("die:0:5", 5)])
def test_bench_launcher_two_rank_rehearsal_exit_codes(spec, want_rc):
    """`python bench.py --gpus 2` without a launcher starts its own ranks (bench.spawn_ranks).  CPU rehearsal over gloo
    (GLSDET_BENCH_STUB): all ranks leave cleanly -> 0 and rank 0's line on stdout; one rank dies while the other waits in a
    collective -> the launcher stops the survivor and returns the dead rank's code instead of hanging."""
    import time
    env = dict(os.environ, GLSDET_BENCH_STUB=spec)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=240)
    assert p.returncode == want_rc, (p.returncode, p.stderr.decode()[-2000:])
    assert time.time() - t0 < 120
    if want_rc == 0:
        import json
        assert json.loads(p.stdout.decode().strip().splitlines()[-1]) == {"stub": "ok", "n_gpus": 2}
