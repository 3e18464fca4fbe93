"""Optional adapter that plugs the drop-in classes into the reference's mmdet / mmdet3d registries so that
`configs/preworld/**.py` build them unmodified ("B-module").

    import preworld_amd.registry as R
    R.register_into_mmdet3d()        # after `import mmdet3d.models`

Every `type` the six PreWorld configs put on the camera -> occupancy path is covered (tests/test_registry_cpu.py builds
all six resolved `model` dicts, tests/golden/preworld_configs.json): the view transformer the configs really name
(`LSSViewTransformerBEVStereo`, and its bases), `CustomResNet3D`, `LSSFPN3D`, `OccHead`, `NerfHead`,
`CustomFocalLoss`, and the detectors `BEVStereo4DOCC` / `PreWorld` / `PreWorld4DTraj` -- the forecasting recursion,
`final_conv` and the attribute MLPs are attributes of the detector, so they can only be swapped by re-registering the
detector.

The reference registers its classes in DIFFERENT registries (view transformers in mmdet3d.models.builder.NECKS,
view_transformer.py:12-15,807; CustomResNet3D in mmdet.models.BACKBONES, resnet.py:7,126; LSSFPN3D / FPN_LSS in
mmdet.models.NECKS, lss_fpn.py:9,12,103; OccHead / NerfHead in mmdet.models.HEADS, occupancy_head.py:13,45,
nerf_head.py:103-104; detectors in mmdet.models.DETECTORS, preworld.py:6,23).  Re-registration needs `force=True`
(mmcv 1.6.0 Registry raises KeyError on duplicates).  mmcv / mmdet are not installed in this image, so the adapter takes
the registry objects as arguments and is unit-tested against a minimal stand-in with the same
`register_module(name=None, force=False, module=None)` signature.

The drop-ins are INFERENCE classes (eval-mode BatchNorm folded into the convs, no conv backward): `register()` therefore
defaults to `inference_only=True` semantics -- it replaces the classes for `tools/test*.py` runs; for a training run call
`register(..., training=True)`, which swaps only classes whose REGISTRY_OF entry says they have a backward on the HIP kernels --
as of round 2 all of them: view transformers, `CustomResNet3D`, `LSSFPN3D`, `OccHead`, `NerfHead`, `CustomFocalLoss` and the three
detectors with their `forward_train` (csrc/pw_train.hip, preworld_amd/train.py)."""
from . import builder

# reference type name -> (registry, can it train?)   -- "can train" = forward under autograd + backward on HIP kernels, pinned by
# gradient fixtures of the imported reference module (tests/test_gpu_train.py, test_gpu_render.py, test_gpu_lss.py)
REGISTRY_OF = {
    'LSSViewTransformer': ('mmdet3d.NECKS', True),            # pooling through ops.bev_pool_v2 (pw_bev_pool_v2_backward)
    'LSSViewTransformerBEVDepth': ('mmdet3d.NECKS', True),    # + DepthNet (PyTorch) and get_depth_loss
    'LSSViewTransformerBEVStereo': ('mmdet3d.NECKS', True),
    'CustomResNet3D': ('mmdet.BACKBONES', True),              # batch-stat BN + conv dgrad / wgrad (preworld_amd/train.py)
    'LSSFPN3D': ('mmdet.NECKS', True),                        # per-level 1x1x1 conv + trilinear up-sampling adjoint + BN
    'OccHead': ('mmdet.HEADS', True),                         # conv / BN on HIP kernels, the 16 -> 8 -> 18 layers as library GEMMs
    'NerfHead': ('mmdet.HEADS', True),                        # fused render forward + backward (ops.RenderRays)
    'CustomFocalLoss': ('mmdet.LOSSES', True),
    'BEVStereo4DOCC': ('mmdet.DETECTORS', True),              # forward_train: depth loss + softmax CE on the predicter MLP
    'PreWorld': ('mmdet.DETECTORS', True),                    # forward_train of the fine-tune / pre-train configs
    'PreWorld4DTraj': ('mmdet.DETECTORS', True),              # temporal forward_train: forecast steps, trajectory branch, per-state losses
}


def replacements():
    """{type name: (registry key, class)} -- the image-side stand-ins (SwinTransformer, FPN_LSS) are NOT registered.  Note that
    the drop-in DETECTORS build their sub-modules through preworld_amd.builder (its own type table), so a detector built from
    a config gets the plain-PyTorch SwinTransformer / FPN_LSS of image_encoder.py (same state-dict keys as the reference's, so
    the same checkpoint loads); to keep the reference's own image modules, pass them as already-built nn.Modules
    (builder.build returns non-dict arguments unchanged)."""
    tab = builder.table()
    return {name: (key, tab[name]) for name, (key, _) in REGISTRY_OF.items()}


REPLACEMENTS = None     # filled lazily (builder.table() imports every module of the package)


def register(registries, training=False):
    """registries: dict with keys 'mmdet3d.NECKS', 'mmdet.BACKBONES', 'mmdet.NECKS', 'mmdet.HEADS', 'mmdet.DETECTORS'
    [, 'mmdet.LOSSES'] mapping to mmcv-style Registry objects (missing keys are skipped).  Returns the list of
    (registry key, type name) that were replaced."""
    done = []
    for name, (key, cls) in replacements().items():
        if key not in registries:
            continue
        if training and not REGISTRY_OF[name][1]:
            continue
        registries[key].register_module(name=name, force=True, module=cls)
        done.append((key, name))
    return done


def register_into_mmdet3d(training=False):
    """Resolve the real registries (needs mmdet3d's dependencies importable) and register."""
    from mmdet.models import BACKBONES, DETECTORS, HEADS, NECKS as MMDET_NECKS
    from mmdet.models.builder import LOSSES
    from mmdet3d.models.builder import NECKS as MMDET3D_NECKS
    return register({'mmdet3d.NECKS': MMDET3D_NECKS, 'mmdet.BACKBONES': BACKBONES, 'mmdet.NECKS': MMDET_NECKS,
                     'mmdet.HEADS': HEADS, 'mmdet.DETECTORS': DETECTORS, 'mmdet.LOSSES': LOSSES}, training=training)
