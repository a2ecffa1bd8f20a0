"""CPU tests of the product package's host logic: config / registry surface,
state_dict key parity with the reference (tables captured in the goldens),
Swin backbone (plain torch, no HIP op inside) against the reference golden,
C-ABI library loads and exports every declared symbol, oracle isolation."""
import ctypes
import os
import re

import pytest
import torch

import common as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = os.path.join(ROOT, "partdistillation_amd", "configs")


def test_capi_library_loads_and_exports_all_declared_symbols():
    from partdistillation_amd import lib
    lib.build()
    declared = set()
    for h in os.listdir(os.path.join(ROOT, "include")):
        text = open(os.path.join(ROOT, "include", h)).read()
        declared |= set(re.findall(r"\b(pd_[a-z0-9_]+)\s*\(", text))
    assert {"pd_msda_forward", "pd_msda_backward", "pd_lsa_batched", "pd_last_error", "pd_abi_version"} <= declared
    so = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(so, name), f"libpd_hip.so does not export {name}"
    assert declared <= set(lib.SIGNATURES)
    assert lib.load().pd_abi_version() == lib.ABI_VERSION


def test_product_never_imports_oracle():
    """the oracle is test infrastructure: nothing under partdistillation_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "partdistillation_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f"{f} imports the oracle"
                assert "/root/reference" not in text, f"{f} reads the reference at run time"


def test_ops_fail_loudly_without_gpu_tensors():
    import partdistillation_amd.MultiScaleDeformableAttention as MSDA
    from partdistillation_amd.functions import lsa
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(torch.zeros(1, 4, 1, 2), torch.tensor([[2, 2]]), torch.tensor([0]),
                                    torch.zeros(1, 1, 1, 1, 1, 2), torch.zeros(1, 1, 1, 1, 1), 2)
    with pytest.raises(RuntimeError, match="GPU only"):
        lsa.solve_batched(torch.zeros(1, 4, 2), torch.tensor([2]))


def _cfg(extra=()):
    from partdistillation_amd.config import setup_cfg
    return setup_cfg(os.path.join(CONFIGS, "proposal_learning", "r50_mask2former.yaml"), ["MODEL.DEVICE", "cpu"] + list(extra))


def test_registries_and_config_surface():
    import partdistillation_amd.modeling  # noqa: F401  (registers)
    import partdistillation_amd.proposal_model  # noqa: F401
    import partdistillation_amd.part_distillation_model  # noqa: F401
    from partdistillation_amd.compat import (BACKBONE_REGISTRY, META_ARCH_REGISTRY, SEM_SEG_HEADS_REGISTRY,
                                             TRANSFORMER_DECODER_REGISTRY)
    for reg, names in ((META_ARCH_REGISTRY, ["ProposalModel", "PartDistillationModel"]),
                       (BACKBONE_REGISTRY, ["build_resnet_backbone", "D2SwinTransformer"]),
                       (SEM_SEG_HEADS_REGISTRY, ["MaskFormerHead", "MSDeformAttnPixelDecoder"]),
                       (TRANSFORMER_DECODER_REGISTRY, ["MultiScaleMaskedTransformerDecoder", "PartDistillationTransformerDecoder"])):
        for n in names:
            assert n in reg
    cfg = _cfg()
    assert cfg.MODEL.META_ARCHITECTURE == "ProposalModel" and cfg.MODEL.MASK_FORMER.DEC_LAYERS == 10
    assert cfg.MODEL.MASK_FORMER.TRAIN_NUM_POINTS == 12544 and cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE == 0.01
    with pytest.raises(KeyError):
        cfg.merge_from_list(["CUSTOM_DATASETS.MIN_OBJECT_AREA_RATIO", "0.1"])   # unknown key fails
    for f in ("proposal_learning/swinl_mask2former.yaml", "part_distillation/swinb_mask2former.yaml",
              "part_distillation/swinl_mask2former.yaml", "part_distillation/swinl_mask2former_fp8.yaml"):
        from partdistillation_amd.config import setup_cfg
        c = setup_cfg(os.path.join(CONFIGS, f))
        assert c.MODEL.BACKBONE.NAME == "D2SwinTransformer"
        if f.endswith("_fp8.yaml"):                          # BASELINE config 5 as named: Swin-L, 1280^2, fp8 GEMMs in the backbone
            assert c.MODEL.SWIN.FP8_GEMM is True and c.MODEL.SWIN.FP8_MIN_K == 384 and c.INPUT.IMAGE_SIZE == 1280
            assert c.MODEL.SWIN.EMBED_DIM == 192 and c.MODEL.META_ARCHITECTURE == "PartDistillationModel"


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="reference only exists in the build container")
def test_reference_yamls_load_unchanged():
    import glob
    from partdistillation_amd.config import setup_cfg
    files = [f for f in glob.glob("/root/reference/configs/**/*.yaml", recursive=True) if "/detic/" not in f]
    assert len(files) >= 15
    for f in files:
        setup_cfg(f)


def test_state_dict_keys_match_reference(golden):
    """pixel decoder + predictor key names / shapes == tables dumped from the reference modules (C1 dims)."""
    from partdistillation_amd.compat import META_ARCH_REGISTRY
    import partdistillation_amd.modeling, partdistillation_amd.proposal_model  # noqa: F401,E401
    g = golden("head_c1")
    model = META_ARCH_REGISTRY.get("ProposalModel")(_cfg())
    sd = model.state_dict()
    for prefix, table in (("sem_seg_head.pixel_decoder.", g["table_pd"]), ("sem_seg_head.predictor.", g["table_dec"])):
        ours = {k[len(prefix):]: (tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items() if k.startswith(prefix)}
        assert ours == {k: (tuple(s), d) for k, (s, d) in table.items()}
    assert "criterion.empty_weight" in sd and "pixel_mean" not in sd
    bb = [k for k in sd if k.startswith("backbone.")]
    assert "backbone.stem.conv1.weight" in bb and "backbone.res5.2.conv3.norm.running_var" in bb
    assert "backbone.res2.0.shortcut.weight" in bb and "backbone.res2.1.shortcut.weight" not in bb


def test_swin_matches_reference_golden(golden):
    from partdistillation_amd.modeling.backbone.swin import SwinTransformer
    g = golden("swin_tiny")
    cfg = C.SWIN_TINY
    net = SwinTransformer(pretrain_img_size=cfg["pretrain_img_size"], patch_size=cfg["patch_size"],
                          embed_dim=cfg["embed_dim"], depths=list(cfg["depths"]), num_heads=list(cfg["num_heads"]),
                          window_size=cfg["window_size"], drop_path_rate=0.0)
    ours = C.table_of(net.state_dict())
    assert ours == {k: (tuple(s), d) for k, (s, d) in g["table"].items()}
    net.load_state_dict(C.seeded_weights(g["table"], 103), strict=False)
    x = C.seeded((cfg["batch"], 3, *cfg["image"]), 901).requires_grad_()
    outs = net(x)
    for k, d in g["outs"].items():
        C.check_digest(outs[k], d, 1e-4, 1e-5, k)
    loss = sum((v * C.seeded(v.shape, 910 + i)).sum() for i, (k, v) in enumerate(sorted(outs.items())))
    torch.testing.assert_close(loss.double(), g["loss"], rtol=1e-5, atol=1e-3)
    loss.backward()
    named = dict(net.named_parameters())
    for k, d in g["grads"].items():
        C.check_digest(named[k].grad, d, 2e-3, 1e-4, "grad " + k)
    C.check_digest(x.grad, g["grad_x"], 2e-3, 1e-4, "grad x")


def test_pinned_ring_hands_out_distinct_staging_buffers():
    """host tables that are rewritten every step must not be staged through ONE reused buffer: the async copy of step n
    may still be pending when step n+1 is issued (that race produced wrong gradient addresses when the host ran ahead)"""
    import torch
    from partdistillation_amd.functions.fused import GatherPlan, PinnedRing
    ring = PinnedRing(4, torch.int64, pin=False, slots=16)     # two blocks of 8 slots, one completion event per block
    seen = []
    for i in range(32):
        buf = ring.acquire()
        buf.fill_(i)
        ring.release()
        seen.append(buf.data_ptr())
    assert len(set(seen[:16])) == 16 and seen[:16] == seen[16:]
    plan = GatherPlan([5, 3], [0, 8], torch.device("cpu"))
    a, b = torch.arange(5.0), torch.ones(3)
    plan.upload([a, b])
    assert plan.src_ptrs.tolist() == [a.data_ptr(), b.data_ptr()]
    plan.upload([None], t_begin=1)
    assert plan.src_ptrs.tolist() == [a.data_ptr(), 0]


def test_proposal_generation_model_registers_and_builds():
    """BASELINE config 4 meta-architecture behind the reference's registry name / config keys; its label-map op is
    GPU-only and fails loudly on the CPU"""
    import torch
    import partdistillation_amd.modeling  # noqa: F401
    import partdistillation_amd.proposal_generation_model as pg
    from partdistillation_amd.compat import META_ARCH_REGISTRY
    from partdistillation_amd.config import setup_cfg
    assert "ProposalGenerationModel" in META_ARCH_REGISTRY
    cfg = setup_cfg(os.path.join(CONFIGS, "proposal_generation", "r50.yaml"))
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg).eval()
    assert isinstance(model, pg.ProposalGenerationModel) and model.backbone_feature_key_list == ["res3", "res4"]
    assert model.distance_metric == "dot" and model.num_superpixel_clusters == 4 and not model.feature_normalize
    setup_cfg(os.path.join(CONFIGS, "proposal_generation", "swinl.yaml"))
    with pytest.raises(RuntimeError, match="GPU only"):
        model._label_map(torch.zeros(4, 4, 4), torch.ones(32, 32, dtype=torch.bool), (32, 32), (32, 32), 32, 32)
    assert model.training is False
    with pytest.raises(AssertionError):
        model.train()([])


@pytest.mark.parametrize("tag,freeze", [("full", []), ("frozen", ["backbone", "encoder"])])
def test_optimizer_param_groups_match_reference_build_optimizer(golden, tag, freeze):
    """engine.optimizer.param_hyperparams against the table the REAL BaseTrainer.build_optimizer (base_trainer.py:64-148)
    produced for the same module tree (Swin + MaskFormerHead with the part-distillation decoder; golden `optimizer`):
    per-parameter lr (backbone x0.1) and weight decay (0 for norms, embeddings, relative-position tables), and the set
    of parameters FREEZE_KEYS turns off (full fine-tune / backbone + encoder frozen as the shipped scripts do)."""
    import types
    from partdistillation_amd.engine.optimizer import param_hyperparams
    from partdistillation_amd.modeling.backbone.swin import SwinTransformer
    from partdistillation_amd.modeling.meta_arch.mask_former_head import MaskFormerHead
    from partdistillation_amd.modeling.pixel_decoder.msdeformattn import MSDeformAttnPixelDecoder
    from partdistillation_amd.modeling.transformer_decoder.part_distillation_transformer_decoder import PartDistillationTransformerDecoder
    from partdistillation_amd.compat import ShapeSpec
    g = golden("optimizer")[tag]
    cfg, sw = C.TINY, C.SWIN_TINY
    model = torch.nn.Module()
    model.backbone = SwinTransformer(pretrain_img_size=sw["pretrain_img_size"], patch_size=sw["patch_size"], embed_dim=sw["embed_dim"],
                                     depths=list(sw["depths"]), num_heads=list(sw["num_heads"]), window_size=sw["window_size"], drop_path_rate=0.0)
    shapes = {f"res{i + 2}": ShapeSpec(channels=sw["embed_dim"] * 2 ** i, stride=s) for i, s in enumerate((4, 8, 16, 32))}
    pdec = MSDeformAttnPixelDecoder(shapes, transformer_dropout=0.0, transformer_nheads=cfg["nheads"], transformer_dim_feedforward=cfg["enc_ffn"],
                                    transformer_enc_layers=cfg["enc_layers"], conv_dim=cfg["conv_dim"], mask_dim=cfg["mask_dim"], norm="GN",
                                    transformer_in_features=["res3", "res4", "res5"], common_stride=4)
    dec = PartDistillationTransformerDecoder(cfg["conv_dim"], True, num_object_classes=5, num_part_classes=4, num_classes=cfg["num_classes"],
                                             hidden_dim=cfg["conv_dim"], num_queries=cfg["queries"], nheads=cfg["nheads"],
                                             dim_feedforward=cfg["dec_ffn"], dec_layers=cfg["dec_layers"], pre_norm=False, mask_dim=cfg["mask_dim"],
                                             enforce_input_project=False, query_feature_normalize=False)
    model.sem_seg_head = MaskFormerHead(shapes, num_classes=1, pixel_decoder=pdec, transformer_predictor=dec,
                                        transformer_in_feature="multi_scale_pixel_decoder")
    ns = types.SimpleNamespace
    c = ns(SOLVER=ns(WEIGHT_DECAY_NORM=0.0, WEIGHT_DECAY_EMBED=0.0, BASE_LR=1e-4, WEIGHT_DECAY=0.05, BACKBONE_MULTIPLIER=0.1),
           MODEL=ns(MASK_FORMER=ns(FREEZE_KEYS=freeze)))
    entries = param_hyperparams(c, model)
    ours = {e["name"]: (e["lr"], e["weight_decay"]) for e in entries}
    assert set(ours) == set(g["table"]), (sorted(set(ours) ^ set(g["table"]))[:10])
    for k, v in g["table"].items():
        assert abs(ours[k][0] - float(v[0])) < 1e-15 and abs(ours[k][1] - float(v[1])) < 1e-15, (k, ours[k], v.tolist())
    assert sorted(n for n, p in model.named_parameters() if not p.requires_grad) == g["frozen"]
    assert g["optimizer_class"] == "FullModelGradientClippingOptimizer" and g["base_class"] == "AdamW"


def test_msda_is_a_registered_torch_library_operator():
    """the operator the reference exposes through pybind (ops/src/vision.cpp:19-22) is a torch.library custom op
    here — schema, fake implementation (shape inference without a GPU) and autograd registration; only a CUDA kernel exists, so a
    CPU call raises like the reference's AT_ERROR("Not implemented on the CPU") instead of falling back to anything."""
    import partdistillation_amd.MultiScaleDeformableAttention as MSDA  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    fwd, bwd = torch.ops.pd.ms_deform_attn_forward.default, torch.ops.pd.ms_deform_attn_backward.default
    assert str(fwd._schema).startswith("pd::ms_deform_attn_forward(Tensor value, Tensor spatial_shapes, Tensor level_start_index, "
                                       "Tensor sampling_loc, Tensor attn_weight, SymInt im2col_step) -> Tensor")
    assert "-> (Tensor, Tensor, Tensor)" in str(bwd._schema)
    with FakeTensorMode():
        v = torch.empty(2, 84, 8, 32, device="cuda")
        sh, lv = torch.empty(3, 2, dtype=torch.long, device="cuda"), torch.empty(3, dtype=torch.long, device="cuda")
        loc, at = torch.empty(2, 50, 8, 3, 4, 2, device="cuda"), torch.empty(2, 50, 8, 3, 4, device="cuda")
        out = torch.ops.pd.ms_deform_attn_forward(v, sh, lv, loc, at, 128)
        assert tuple(out.shape) == (2, 50, 256) and out.device.type == "cuda"
        gv, gl, ga = torch.ops.pd.ms_deform_attn_backward(v, sh, lv, loc, at, out, 128)
        assert gv.shape == v.shape and gl.shape == loc.shape and ga.shape == at.shape
    with pytest.raises(NotImplementedError):
        torch.ops.pd.ms_deform_attn_forward(torch.zeros(1, 4, 2, 2), torch.tensor([[2, 2]]), torch.tensor([0]), torch.zeros(1, 3, 2, 1, 2, 2),
                                            torch.zeros(1, 3, 2, 1, 2), 1)


def test_cat_levels_is_torch_cat_with_view_gradients():
    """modeling/pixel_decoder/msdeformattn.CatLevels: the encoder's level tokens side by side; same values and gradients as
    torch.cat of the flattened maps, and the gradients are views of the incoming gradient (no copies)."""
    from partdistillation_amd.modeling.pixel_decoder.msdeformattn import CatLevels
    torch.manual_seed(0)
    srcs = [torch.randn(2, 8, h, w, requires_grad=True) for h, w in [(2, 3), (4, 6), (8, 12)]]
    ref = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    out = CatLevels.apply(*srcs)
    assert torch.equal(out, ref)
    g = torch.randn_like(out)
    for a, b in zip(torch.autograd.grad(ref, srcs, g), torch.autograd.grad(out, srcs, g)):
        assert torch.equal(a, b)
        assert b.untyped_storage().data_ptr() == g.untyped_storage().data_ptr()


def test_pseudo_targets_form_object_masks_on_first_read():
    from partdistillation_amd.proposal_model import _PseudoTargets
    m = torch.rand(3, 5, 7) > 0.5
    t = _PseudoTargets({"labels": torch.zeros(3, dtype=torch.long), "masks": m})
    assert "object_masks" not in t
    om = t["object_masks"]
    assert om.dtype == torch.int64 and torch.equal(om, m.sum(0, keepdim=True)) and "object_masks" in t
    with pytest.raises(KeyError):
        t["nothing"]
