"""`type`-keyed construction of the hot-path components, standing in for mmdet3d/models/builder.py:16-68
(`build_neck` / `build_backbone` / `build_head` / `build_loss`) when mmcv / mmdet are not installed (they are not in
this image).  The table maps every `type` the six `configs/preworld/**.py` files use on the camera -> occupancy path to
the class of this package that replaces it; preworld_amd.registry force-registers the same table into the real
mmdet / mmdet3d registries when those are importable."""


def table():
    from . import detectors, image_encoder, losses, modules
    return {
        # mmdet3d NECKS (view_transformer.py:15,702,807)
        'LSSViewTransformer': modules.LSSViewTransformer,
        'LSSViewTransformerBEVDepth': modules.LSSViewTransformerBEVDepth,
        'LSSViewTransformerBEVStereo': modules.LSSViewTransformerBEVStereo,
        # mmdet BACKBONES / NECKS / HEADS / LOSSES (resnet.py:126, lss_fpn.py:12,103, occupancy_head.py:45, nerf_head.py:104)
        'CustomResNet3D': modules.CustomResNet3D,
        'LSSFPN3D': modules.LSSFPN3D,
        'OccHead': modules.OccHead,
        'NerfHead': modules.NerfHead,
        'CustomFocalLoss': losses.CustomFocalLoss,
        # image side, plain PyTorch-ROCm (swin.py:680, lss_fpn.py:12)
        'SwinTransformer': image_encoder.SwinTransformer,
        'FPN_LSS': image_encoder.FPN_LSS,
        # mmdet DETECTORS (bevdet_occ.py:45, preworld.py:23, preworld_temporal_traj.py:26)
        'BEVStereo4DOCC': detectors.BEVStereo4DOCC,
        'PreWorld': detectors.PreWorld,
        'PreWorld4DTraj': detectors.PreWorld4DTraj,
    }


def build(cfg, default_type=None):
    """cfg: dict(type=..., **kwargs) as written in the reference configs (or an already built nn.Module / None)."""
    if cfg is None or not isinstance(cfg, dict):
        return cfg
    cfg = dict(cfg)
    name = cfg.pop('type', default_type)
    tab = table()
    if name not in tab:
        raise KeyError('no preworld_amd replacement for type=%r (known: %s)' % (name, sorted(tab)))
    return tab[name](**cfg)


build_neck = build_backbone = build_head = build_loss = build_detector = build_model = build
