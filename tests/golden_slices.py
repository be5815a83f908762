"""Which slices of which gradients the full-geometry fixtures keep (shared by tests/golden/make_golden.py and the tests;
the tensors themselves are 0.4-92 M elements, the fixtures hold norms + these slices)."""
S = slice
B16_GRAD_SLICES = {'ln_pre.weight': None, 'class_embedding': None, 'transformer.resblocks.0.attn.in_proj_bias': None,
                   'transformer.resblocks.0.mlp.c_fc.bias': None, 'transformer.resblocks.0.ln_2.weight': None,
                   'positional_embedding': (S(None, None, 24), S(None)), 'proj': (S(None, None, 8), S(None, None, 8)),
                   'transformer.resblocks.0.attn.in_proj_weight': (S(None, None, 16), S(None, None, 16)),
                   'transformer.resblocks.0.mlp.c_proj.weight': (S(None, None, 8), S(None, None, 32)),
                   'conv1.weight': (S(None, None, 8), S(None), S(None, None, 4), S(None, None, 4))}
FULL_STEP_GRAD_SLICES = {
    'temp': None, 'queries': None, 'text_proj.weight': (S(None, None, 4), S(None, None, 8)),
    'vision_proj.weight': (S(None, None, 4), S(None, None, 8)), 'caption_proj1.bias': None, 'cproj.bias': None,
    'clip_encoder.visual.ln_post.weight': None, 'clip_encoder.visual.transformer.resblocks.0.ln_1.bias': None,
    'clip_encoder.visual.transformer.resblocks.11.mlp.c_fc.bias': None,
    'clip_encoder.visual.transformer.resblocks.5.attn.in_proj_weight': (S(None, None, 16), S(None, None, 16)),
    'clip_encoder.visual.conv1.weight': (S(None, None, 8), S(None), S(None, None, 4), S(None, None, 4)),
    'text_encoder.encoder.layer.0.attention.self.query.bias': None, 'text_encoder.encoder.layer.11.output.LayerNorm.weight': None,
    'text_encoder.encoder.layer.6.intermediate.dense.weight': (S(None, None, 32), S(None, None, 8)),
    'text_encoder.embeddings.position_embeddings.weight': (S(0, 12), S(None, None, 8)),
    'crossattn.layers.1.linear2.weight': (S(None, None, 8), S(None, None, 8)),
    'crossattn_query.layers.0.multihead_attn.in_proj_bias': None, 'crossattn2.layers.0.norm2.weight': None}
