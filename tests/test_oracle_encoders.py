"""CPU checks of the oracle's restated third-party encoders (parity otherwise unpinned)."""
import torch

from oracle.encoders import ResNetFeatureExtractor, RobertaEncoder


def test_resnet_shapes_and_names():
    m = ResNetFeatureExtractor((1, 2, 1, 1), width=8).eval()
    y = m(torch.randn(2, 3, 224, 224))
    assert y.shape == (2, 8 * 8 * 4, 7, 7)                      # tell/models/resnet.py:107 ([B,2048,7,7] at width 64)
    keys = set(m.state_dict())
    for k in ['conv1.weight', 'bn1.running_mean', 'layer1.0.conv2.weight', 'layer1.0.downsample.0.weight',
              'layer2.0.downsample.1.running_var', 'layer2.1.bn3.weight', 'fc.weight']:
        assert k in keys, k
    full = ResNetFeatureExtractor()
    n = sum(p.numel() for p in full.parameters())
    assert n == 60192808, n                                      # torchvision resnet152 parameter count


def test_roberta_matches_published_architecture():
    """oracle RoBERTa == transformers.RobertaModel (independent implementation of the same
    published architecture) on the non-padded positions, all hidden states."""
    from transformers import RobertaConfig, RobertaModel
    torch.manual_seed(0)
    V, E, FF, L, H, P = 120, 64, 128, 3, 4, 40
    ora = RobertaEncoder(vocab=V, dim=E, ffn=FF, layers=L, heads=H, max_positions=P).eval()
    cfg = RobertaConfig(vocab_size=V, hidden_size=E, intermediate_size=FF, num_hidden_layers=L,
                        num_attention_heads=H, max_position_embeddings=P + 2, pad_token_id=1, type_vocab_size=1,
                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=1e-5,
                        hidden_act='gelu')
    hf = RobertaModel(cfg, add_pooling_layer=False).eval()
    enc = ora.model.decoder.sentence_encoder
    sd = {}
    sd['embeddings.word_embeddings.weight'] = enc.embed_tokens.weight
    sd['embeddings.position_embeddings.weight'] = enc.embed_positions.weight
    sd['embeddings.token_type_embeddings.weight'] = torch.zeros(1, E)
    sd['embeddings.LayerNorm.weight'] = enc.emb_layer_norm.weight
    sd['embeddings.LayerNorm.bias'] = enc.emb_layer_norm.bias
    for i, l in enumerate(enc.layers):
        p = 'encoder.layer.%d.' % i
        w, b = l.self_attn.in_proj_weight, l.self_attn.in_proj_bias
        for j, n in enumerate(['query', 'key', 'value']):
            sd[p + 'attention.self.%s.weight' % n] = w[j * E:(j + 1) * E]
            sd[p + 'attention.self.%s.bias' % n] = b[j * E:(j + 1) * E]
        sd[p + 'attention.output.dense.weight'] = l.self_attn.out_proj.weight
        sd[p + 'attention.output.dense.bias'] = l.self_attn.out_proj.bias
        sd[p + 'attention.output.LayerNorm.weight'] = l.self_attn_layer_norm.weight
        sd[p + 'attention.output.LayerNorm.bias'] = l.self_attn_layer_norm.bias
        sd[p + 'intermediate.dense.weight'] = l.fc1.weight
        sd[p + 'intermediate.dense.bias'] = l.fc1.bias
        sd[p + 'output.dense.weight'] = l.fc2.weight
        sd[p + 'output.dense.bias'] = l.fc2.bias
        sd[p + 'output.LayerNorm.weight'] = l.final_layer_norm.weight
        sd[p + 'output.LayerNorm.bias'] = l.final_layer_norm.bias
    missing = hf.load_state_dict({k: v.detach().clone() for k, v in sd.items()}, strict=False)
    assert not [k for k in missing.missing_keys if 'position_ids' not in k], missing
    ids = torch.randint(3, V, (3, 20))
    ids[:, 0] = 0
    ids[1, 14:] = 1
    ids[2, 5:] = 1
    with torch.no_grad():
        mine = ora.extract_features(ids, return_all_hiddens=True)
        ref = hf(input_ids=ids, attention_mask=(ids != 1).long(), output_hidden_states=True).hidden_states
    assert len(mine) == len(ref) == L + 1
    keep = ids != 1
    for a, b in zip(mine, ref):
        torch.testing.assert_close(a[keep], b[keep], rtol=1e-4, atol=1e-5)
