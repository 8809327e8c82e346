"""Drop-in boundary: constructor surface, cfg handling, state_dict names, loud failure off-GPU."""
import ctypes
import os
import re

import pytest
import torch

from rsgnet_amd import as_cfg, hip, make_cfg, models
from tests.util import ROOT, TAGS, case_cfg, manifest


@pytest.mark.parametrize("tag", list(TAGS))
def test_state_dict_matches_reference_manifest(tag):
    m = getattr(models, TAGS[tag][0]).get_pose_net(case_cfg(tag), is_train=False)
    want = manifest(tag)["state_dict"]
    got = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert list(got) == list(want)            # same names, same order
    assert got == want


def test_w48_builds_with_list_valued_num_type_vector():
    cfg = make_cfg("pose_rsgnet", 48, (288, 384))
    raw = {"MODEL": dict(cfg.MODEL)}
    raw["MODEL"]["NUM_TYPE_VECTOR"] = [600]          # the malformed shipped yaml
    m = models.pose_rsgnet.get_pose_net(raw, is_train=False)
    assert m.type_fc[0].weight.shape == (600, 600)
    assert m.relation_head.W[1].num_groups == 8 and m.relation_head.W[1].num_channels == 48


def test_cfg_item_and_attr_access():
    cfg = as_cfg({"MODEL": {"NAME": "pose_hrnet", "EXTRA": make_cfg().MODEL.EXTRA}})
    assert cfg.MODEL.EXTRA.STAGE2.NUM_CHANNELS == cfg["MODEL"]["EXTRA"]["STAGE2"]["NUM_CHANNELS"] == [32, 64]
    assert cfg.MODEL.NUM_JOINTS == 17 and cfg.MODEL.UDP_POSE_ON is False


def test_branch_mismatch_raises_value_error():
    cfg = make_cfg()
    cfg.MODEL.EXTRA.STAGE3.NUM_BLOCKS = [4, 4]
    with pytest.raises(ValueError):
        models.pose_hrnet.get_pose_net(cfg, False)


def test_missing_pretrained_raises_value_error(tmp_path):
    cfg = make_cfg()
    cfg.MODEL.PRETRAINED = str(tmp_path / "nope.pth")
    with pytest.raises(ValueError):
        models.pose_hrnet.get_pose_net(cfg, is_train=True)


def test_init_weights_partial_load(tmp_path):
    cfg = make_cfg("pose_rsgnet")
    src = models.pose_rsgnet.get_pose_net(cfg, False)
    path = tmp_path / "w.pth"
    torch.save(src.state_dict(), path)
    cfg.MODEL.PRETRAINED = str(path)
    m = models.pose_rsgnet.get_pose_net(cfg, is_train=True)
    assert torch.equal(m.stage3[1].branches[2][0].conv1.weight, src.stage3[1].branches[2][0].conv1.weight)
    assert not torch.equal(m.vis_conv[0].weight, src.vis_conv[0].weight)     # head is not in PRETRAINED_LAYERS
    assert float(m.vis_conv[0].weight.std()) < 2e-3                           # N(0, 1e-3) init


def test_forward_off_gpu_fails_loudly():
    m = models.pose_hrnet.get_pose_net(make_cfg(), False).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 256, 192))


def test_param_signature_sees_every_way_the_weights_can_change():
    """The check in front of every forward (models/_base.py): cheap (no generator walk over the module tree), and it
    moves on an in-place write to a parameter or a buffer, on a parameter OBJECT swapped for another, on load_state_dict
    and on a FusedAdam step; it stands still otherwise."""
    from rsgnet_amd import optim
    m = models.pose_hrnet.get_pose_net(make_cfg(), False).eval()
    s0 = m._param_signature()
    assert m._param_signature() == s0
    with torch.no_grad():
        m.final_layer.weight.mul_(0.5)
    s1 = m._param_signature()
    assert s1 != s0
    m.bn1.running_mean.add_(1.0)
    s2 = m._param_signature()
    assert s2 != s1
    m.final_layer.weight = torch.nn.Parameter(m.final_layer.weight.detach().clone())     # same values and version, other object
    s3 = m._param_signature()
    assert s3 != s2
    m.load_state_dict(m.state_dict())
    s4 = m._param_signature()
    assert s4 != s3
    optim.PARAM_EPOCH += 1                     # what FusedAdam.step does after its kernel wrote through raw pointers
    assert m._param_signature() != s4
    optim.PARAM_EPOCH -= 1
    assert m._param_signature() == s4
    n_tensors = len(list(m.parameters())) + len(list(m.buffers()))
    assert len(m.__dict__["_sig_tensors"]) == n_tensors          # the cached flat list covers every tensor of the model
    m.bn1.register_buffer("extra_stat", torch.zeros(3))           # a tensor that did not exist when the list was built
    s5 = m._param_signature()
    assert s5 != s4 and len(m.__dict__["_sig_tensors"]) == n_tensors + 1
    m.bn1.extra_stat.add_(1.0)
    assert m._param_signature() != s5


def test_library_exports_every_declared_symbol():
    lib = hip.lib()
    header = open(os.path.join(ROOT, "include", "rsgnet_hip.h")).read()
    declared = set(re.findall(r"\b(rsg_[a-z0-9_]+)\s*\(", header))
    assert declared == set(hip.SYMBOLS), declared ^ set(hip.SYMBOLS)
    raw = ctypes.CDLL(hip.lib_path())
    for name in declared:
        assert getattr(raw, name) is not None
    assert lib.rsg_version() >= 1


def test_entry_points_reject_bad_arguments_before_any_launch():
    """Argument checks of the C ABI run on the host (no GPU needed): a bad call returns non-zero and rsg_last_error says
    why; nothing is launched."""
    lib = hip.lib()
    assert lib.rsg_trp_mse_workspace_bytes(2, 3072) == 2 * (3072 // 32) * 4 and lib.rsg_trp_mse_workspace_bytes(2, 100) == 0
    buf = (ctypes.c_char * 4096)()
    a = ctypes.addressof(buf) & ~15 | 16                    # any aligned non-null address: the call fails before it is read
    assert lib.rsg_trp_relation_mse(a, 32, a, a, None, 2, 3072, 32, 0, None) != 0
    assert b"workspace" in lib.rsg_last_error()
    assert lib.rsg_trp_relation_mse(a, 32, a, a, a, 2, 3000, 32, 0, None) != 0 and b"S=3000" in lib.rsg_last_error()
    assert lib.rsg_trp_relation_mse_dense(a, 32, None, a, a, 2, 3072, 32, 0, None) != 0
    assert lib.rsg_tune(b"conv_spread", -1) != 0 and lib.rsg_tune(b"no_such_key", 1) != 0
    assert lib.rsg_tune(b"conv_spread", 128) == 0


def test_product_package_never_imports_oracle():
    bad = []
    for dp, _d, files in os.walk(os.path.join(ROOT, "rsgnet_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle", src, re.M):
                    bad.append(f)
    assert not bad, bad


def test_build_is_keyed_on_its_sources_and_the_guarded_sources_hold_no_packed_fp32():
    """The library in the tree was built from the sources, flag table and compiler the tree holds now (rsgnet_amd/build.py:
    hash stored beside the .so), and no code object of the library contains a packed fp32 VALU op (v_pk_{fma,mul,add}_f32:
    every source is built with -packed-fp32-ops, DESIGN 9b')."""
    import pytest
    from rsgnet_amd import build as b
    b.build()
    assert not b.stale()
    rec = b.info()
    assert rec["hash"] == b.source_hash() and rec["sources_hash"] == b.source_hash(with_compiler=False)
    assert "-packed-fp32-ops" in rec["common_flags"]
    assert rec["hash"][:16] in b.describe()
    counts = b.audit_packed_fp32()
    if not counts:
        pytest.skip("llvm-objdump / clang-offload-bundler / object files not available")
    assert len(counts) >= 12 and not any(counts.values()), counts
