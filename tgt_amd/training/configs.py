"""Model configurations named by BASELINE.json (reference YAMLs under
configs/pcqm/tgt_at_200m and tgt_agx2_100m; kwargs as
lib/training_schemes/pcqm/tgt_training.py:68-92 hands them to the model)."""


def tgt_at_24l(num_dist_bins=512, dropouts=True, embed_3d_type='gaussian'):
    """TGT-At 24L pretrain model (configs/pcqm/tgt_at_200m/pretrain/tgt_at_tp.yaml)."""
    return dict(
        model_height=24, layer_multiplier=1, upto_hop=32, embed_3d_type=embed_3d_type,
        num_3d_kernels=128, num_dist_bins=num_dist_bins,
        node_width=768, edge_width=256, num_heads=64, activation='gelu', scale_degree=True,
        triplet_heads=16, triplet_type='attention', triplet_dropout=0,
        node_ffn_multiplier=1., edge_ffn_multiplier=1.,
        source_dropout=0.3 if dropouts else 0, drop_path=0.2 if dropouts else 0,
        node_act_dropout=0.1 if dropouts else 0, edge_act_dropout=0.1 if dropouts else 0,
    )


def tgt_agx2_12x2(num_dist_bins=256, dropouts=True, embed_3d_type='gaussian'):
    """TGT-Agx2 12 shared layers x2 (configs/pcqm/tgt_agx2_100m/*)."""
    cfg = tgt_at_24l(num_dist_bins, dropouts, embed_3d_type)
    cfg.update(model_height=12, layer_multiplier=2, triplet_type='aggregate',
               drop_path=0.1 if dropouts else 0)
    return cfg
