"""Optional adapter that plugs the drop-in modules into the reference's mmdet3d registries so
that `configs/preworld/*.py` build them unmodified ("B-module").

    import preworld_amd.registry as R
    R.register_into_mmdet3d()        # after `import mmdet3d.models`

The reference registers its classes in DIFFERENT registries (view transformers in
mmdet3d.models.builder.NECKS, view_transformer.py:12-15; CustomResNet3D in mmdet.models.BACKBONES,
resnet.py:7,126; LSSFPN3D in mmdet.models.NECKS, lss_fpn.py:9,103; OccHead / NerfHead in
mmdet.models.HEADS, occupancy_head.py:13,45, nerf_head.py:103-104).  Re-registration needs
`force=True` (mmcv 1.6.0 Registry raises KeyError on duplicates).  mmcv/mmdet are not installed in
this image, so the adapter takes the registry objects as arguments and is unit-tested against a
minimal stand-in with the same `register_module(name=None, force=False, module=None)` signature.
"""
from . import losses, modules

# reference type name -> (which registry, replacement class)
REPLACEMENTS = {
    'LSSViewTransformer': ('mmdet3d.NECKS', modules.LSSViewTransformer),
    'CustomResNet3D': ('mmdet.BACKBONES', modules.CustomResNet3D),
    'LSSFPN3D': ('mmdet.NECKS', modules.LSSFPN3D),
    'OccHead': ('mmdet.HEADS', modules.OccHead),
    'NerfHead': ('mmdet.HEADS', modules.NerfHead),
    # mmdet3d/models/loss_utils/focal_loss.py:7,162 registers it in mmdet's LOSSES; preworld.py:117 builds it by name
    'CustomFocalLoss': ('mmdet.LOSSES', losses.CustomFocalLoss),
}


def register(registries):
    """registries: dict with keys 'mmdet3d.NECKS', 'mmdet.BACKBONES', 'mmdet.NECKS', 'mmdet.HEADS' [, 'mmdet.LOSSES']
    mapping to mmcv-style Registry objects.  Returns the list of (registry key, type name)."""
    done = []
    for name, (key, cls) in REPLACEMENTS.items():
        if key not in registries:            # callers that only replace modules may leave LOSSES out
            continue
        registries[key].register_module(name=name, force=True, module=cls)
        done.append((key, name))
    return done


def register_into_mmdet3d():
    """Resolve the real registries (needs mmdet3d's dependencies importable) and register."""
    from mmdet.models import BACKBONES, HEADS, NECKS as MMDET_NECKS   # noqa: F401
    from mmdet.models.builder import LOSSES
    from mmdet3d.models.builder import NECKS as MMDET3D_NECKS
    return register({'mmdet3d.NECKS': MMDET3D_NECKS, 'mmdet.BACKBONES': BACKBONES,
                     'mmdet.NECKS': MMDET_NECKS, 'mmdet.HEADS': HEADS, 'mmdet.LOSSES': LOSSES})
