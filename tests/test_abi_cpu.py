"""CPU (no GPU needed): the C-ABI library loads, exports every symbol include/weasal_hip.h
declares, validates arguments before touching the device, and the numpy facades reject bad
input with the reference's error type."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import REPO


def _declared():
    text = open(os.path.join(REPO, "include", "weasal_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ws_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from weasal_amd import _lib
    names = _declared()
    assert len(names) >= 20
    lib = _lib.lib()
    for n in names:
        assert hasattr(lib, n), "libweasal_hip.so does not export %s" % n
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert b"gfx950" in lib.ws_version()
    assert isinstance(lib.ws_device_count(), int)


def test_argument_validation_needs_no_device():
    from weasal_amd import _lib
    lib = _lib.lib()
    null = C.c_void_p(None)
    one = C.c_void_p(16)     # never dereferenced: validation fails first
    # K != 15 -> unsupported
    rc = lib.ws_kpconv_gather_fwd(one, 4, one, 4, one, 3, one, 8, one, 7, null, null, 1.0, 0, 0, null, one, null, null)
    assert rc == 2 and b"num_kernel_points" in lib.ws_last_error()
    # bad extent
    rc = lib.ws_kpconv_gather_fwd(one, 4, one, 4, one, 3, one, 8, one, 15, null, null, 0.0, 0, 0, null, one, null, null)
    assert rc == 1 and b"KP_extent" in lib.ws_last_error()
    # unknown influence
    rc = lib.ws_kpconv_gather_fwd(one, 4, one, 4, one, 3, one, 8, one, 15, null, null, 1.0, 9, 0, null, one, null, null)
    assert rc == 1
    # empty query set is a no-op
    rc = lib.ws_kpconv_gather_fwd(null, 0, one, 4, one, 3, one, 8, one, 15, null, null, 1.0, 0, 0, null, one, null, null)
    assert rc == 0
    assert lib.ws_transpose_scratch_bytes(1000, 10, 500) > 500 * 4
    with pytest.raises(_lib.WeasalHipError):
        _lib.check(1)


def test_missing_library_fails_loudly(monkeypatch):
    from weasal_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libweasal_hip.so")
    with pytest.raises(_lib.WeasalHipError, match="no CPU fallback"):
        _lib.lib()


def test_operators_refuse_cpu_tensors():
    import torch
    from weasal_amd import _lib, ops
    x = torch.zeros(4, 8)
    i = torch.zeros(4, 2, dtype=torch.int64)
    with pytest.raises(_lib.WeasalHipError):
        ops.max_pool(x, i)
    with pytest.raises(_lib.WeasalHipError):
        ops.kpconv_gather(x, torch.zeros(4, 3), torch.zeros(4, 3), i, torch.zeros(15, 3), 1.0)
    with pytest.raises(_lib.WeasalHipError):
        ops.radius_neighbors(torch.zeros(4, 3), torch.zeros(4, 3), [4], [4], 1.0)


def test_facade_argument_errors():
    """shape / parsing errors are RuntimeError with the reference's messages (wrapper.cpp:127-171)"""
    from weasal_amd.cpp_wrappers.cpp_neighbors import radius_neighbors as rn
    from weasal_amd.cpp_wrappers.cpp_subsampling import grid_subsampling as gs
    p = np.zeros((5, 3), np.float32)
    with pytest.raises(RuntimeError, match="query.shape"):
        rn.batch_query(np.zeros((5, 2)), p, [5], [5], radius=1.0)
    with pytest.raises(RuntimeError, match="support.shape"):
        rn.batch_query(p, np.zeros(5), [5], [5], radius=1.0)
    with pytest.raises(RuntimeError, match="different for queries and supports"):
        rn.batch_query(p, p, [5], [2, 3], radius=1.0)
    with pytest.raises(TypeError):
        rn.batch_query(p, p, [5], [5], 1.0)          # radius is keyword-only ("OOOO|$f")
    with pytest.raises(RuntimeError, match="points.shape"):
        gs.subsample_batch(np.zeros((5, 4)), [5], sampleDl=0.1)
    with pytest.raises(RuntimeError, match="method"):
        gs.subsample(p, sampleDl=0.1, method="median")
    with pytest.raises(RuntimeError, match="features.shape"):
        gs.subsample(p, features=np.zeros((4, 2)), sampleDl=0.1)
    with pytest.raises(RuntimeError, match="classes.shape"):
        gs.subsample_batch(p, [5], classes=np.zeros((4,)), sampleDl=0.1)


def test_config_derived_fields():
    from weasal_amd.config import DALESPLConfig, Config
    c = DALESPLConfig()
    assert c.num_layers == 5 and c.deform_layers == [False] * 5

    class D(Config):
        architecture = ['simple', 'resnetb_deformable', 'resnetb_deformable_strided', 'resnetb', 'nearest_upsample', 'unary']
    d = D()
    assert d.num_layers == 2 and d.deform_layers == [True, False]


def test_kpfcnn_state_dict_keys_match_reference():
    """the reference checkpoint layout (golden g8 holds the reference's state_dict)"""
    from conftest import golden
    from test_oracle_cpu_kpconv import _small_config
    from weasal_amd.architectures import KPFCNN
    g8 = golden("g8_kpfcnn.npz")
    np.random.seed(0)
    net = KPFCNN(_small_config(), np.arange(9), [])
    ours = {k for k in net.state_dict() if "num_batches_tracked" not in k}
    ref = {k[4:] for k in g8.files if k.startswith("sd0/")}
    assert ours == ref
    # optimizer split of utils/trainer_PseudoLabel.py:80-81
    assert not [k for k, _ in net.named_parameters() if 'offset' in k]


def test_calibration_percentiles():
    """limits = smallest width leaving 90 % of the rows uncropped (DALES_PseudoLabel.py:1321-1324)"""
    from weasal_amd.calibration import histogram_size, limits_from_histograms
    from weasal_amd.config import DALESPLConfig
    assert histogram_size(DALESPLConfig()) == int(np.ceil(4 / 3 * np.pi * 6.0 ** 3))
    rng = np.random.default_rng(0)
    counts = [rng.integers(5, 60, size=4000), rng.integers(20, 90, size=700)]
    hists = np.vstack([np.bincount(c, minlength=120)[:120] for c in counts])
    lim = limits_from_histograms(hists, 0.9)
    for c, l in zip(counts, lim):
        assert (c <= l).mean() >= 0.9 and (c <= l - 1).mean() < 0.9


def test_batch_limit_controller_converges_on_a_synthetic_sampler():
    """calibration.BatchLimitController (DALES_PseudoLabel.py:1190-1241): with spheres of ~N points a batch of budget L
    holds about L / N spheres; the loop must settle on the budget that gives the target sphere count"""
    import numpy as np
    from weasal_amd.calibration import BatchLimitController
    rng = np.random.default_rng(0)
    ctl = BatchLimitController(target_b=8, batch_limit=1.0, expected_n=20000)
    limit = ctl.batch_limit
    for _ in range(6000):
        sizes = rng.normal(20000, 1500, size=64).clip(5000)
        b = int(np.searchsorted(np.cumsum(sizes), max(limit, 0.0))) + 1     # spheres that fit the point budget
        limit = ctl.update(b)
        if ctl.converged:
            break
    assert ctl.converged and abs(ctl.estim_b - 8) < 0.5
    assert 6 * 20000 < ctl.batch_limit < 10 * 20000


def test_facades_raise_in_a_forked_child_of_a_gpu_parent(monkeypatch):
    """The reference's DataLoader workers are forked after the net went to the GPU (train_DALES_PseudoLabel.py:291-296):
    in such a child the facades must raise the reference's RuntimeError with the remedy, not crash or hang.  No GPU here:
    the 'runtime was live in the parent' condition is simulated (torch.cuda.is_initialized -> True at fork time)."""
    import os
    import torch
    from weasal_amd.cpp_wrappers import _device
    from weasal_amd.cpp_wrappers.cpp_neighbors import radius_neighbors as rn
    from weasal_amd.cpp_wrappers.cpp_subsampling import grid_subsampling as gs
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:                                   # child: report what the facades did, never return into pytest
        os.close(r)
        msgs = []
        p = np.zeros((5, 3), np.float32)
        for call in (lambda: rn.batch_query(p, p, [5], [5], radius=1.0), lambda: gs.subsample_batch(p, [5], sampleDl=0.5)):
            try:
                call()
                msgs.append("no error")
            except RuntimeError as e:
                msgs.append(str(e))
            except BaseException as e:             # noqa: BLE001
                msgs.append("other: %r" % (e,))
        os.write(w, "\n".join(msgs).encode())
        os._exit(0)
    os.close(w)
    data = b""
    while True:
        chunk = os.read(r, 65536)
        if not chunk:
            break
        data += chunk
    os.waitpid(pid, 0)
    msgs = data.decode().split("\n")
    assert len(msgs) == 2
    for m in msgs:
        assert m == _device.FORK_MESSAGE and "spawn" in m
    assert not _device._forked_from_gpu_parent     # the parent itself is unaffected


def test_pyramid_descriptor_mirror_and_schedule():
    """the ctypes mirror of `struct ws_pyramid_desc` has the library's size, and the per-level plan handed to
    ws_pyramid_build is the schedule of datasets/common.py:487-545 (radii double per level, the deformable radius where a
    level holds deformable blocks, pooling everywhere but on the last level)"""
    import ctypes as C
    from weasal_amd import _lib, config as wcfg, pyramid
    assert C.sizeof(pyramid.PyramidDesc) == _lib.lib().ws_pyramid_desc_bytes()
    for cfg in (wcfg.DALESPLConfig(), wcfg.Vaihingen3DPLConfig(), wcfg.Vaihingen3DWLConfig(), wcfg.DALESDeformConfig()):
        levels = pyramid._schedule(cfg, [1] * 8)
        assert len(levels) == cfg.num_layers
        r = cfg.first_subsampling_dl * cfg.conv_radius
        for l, lv in enumerate(levels):
            deform_here = any('deformable' in b for b in _blocks_of_level(cfg, l))
            want = r * cfg.deform_radius / cfg.conv_radius if deform_here else r
            assert lv["conv_on"] and abs(lv["r_conv"] - want) < 1e-12
            assert lv["pool_on"] == (l + 1 < len(levels))
            if lv["pool_on"]:
                assert abs(lv["dl"] - 2 * r / cfg.conv_radius) < 1e-12 and abs(lv["r_up"] - 2 * lv["r_pool"]) < 1e-12
            r *= 2


def _blocks_of_level(cfg, level):
    out, cur, l = [], [], 0
    for b in cfg.architecture:
        if any(t in b for t in ('pool', 'strided', 'global', 'upsample')):
            if l == level:
                return cur
            cur, l = [], l + 1
            if 'global' in b or 'upsample' in b:
                break
        else:
            cur.append(b)
    return out
