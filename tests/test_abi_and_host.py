"""CPU checks: the C-ABI library loads and exports every symbol include/tell_hip.h declares,
host-side logic (schedules, flat-buffer layout, RNG restatement, config shim, registries)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import tell_amd
    protos = tell_amd.hip.parse_header()
    assert len(protos) >= 45
    lib = ctypes.CDLL(tell_amd.hip.LIB_PATH)
    missing = [n for n in protos if not hasattr(lib, n)]
    assert not missing, missing
    bound = tell_amd.hip.lib()                       # binds restype/argtypes for all of them
    assert bound.tell_abi_version() == 2
    assert bound.tell_opt_chunk() == 1024
    assert bound.tell_last_error() is not None
    for must in ['tell_gemm_nt', 'tell_attn_fwd', 'tell_attn_bwd', 'tell_dynconv_fwd', 'tell_dynconv_bwd',
                 'tell_layernorm_fwd', 'tell_ce_fwd', 'tell_adaptive_partition', 'tell_bertadam_step',
                 'tell_adaptive_logprob_argmax', 'tell_im2col', 'tell_bn_stats']:
        assert must in protos


def test_library_options_are_explicit_and_the_ablations_are_not_in_the_shipped_library():
    """include/tell_hip.h tell_set_option: the launchers' choices are named integers set through the ABI; the library reads
    no environment variable (no getenv import, no TELL_* string), and the wrong-result timing ablations of tools/probes/
    (csrc/options.h TELL_PROBE_LIST) are unknown keys in the shipped build - setting them is an error, not a switch."""
    import re
    import subprocess
    import tell_amd
    hip = tell_amd.hip
    lib = hip.lib()
    assert lib.tell_probe_build() == 0
    keys = hip.option_keys()
    assert len(keys) == len(set(keys)) >= 30 and {'gemm_q4', 'q4_dynamic', 'gemm_s64', 'conv_tile', 'dynconv_block'} <= set(keys)
    # the table in the library is the table in csrc/options.h (keys and defaults)
    src = open(os.path.join(ROOT, 'transform-and-tell_amd', 'csrc', 'options.h')).read()
    shipped = src[src.index('#define TELL_OPTION_LIST'):src.index('#define TELL_PROBE_LIST')]
    probes = src[src.index('#define TELL_PROBE_LIST'):src.index('enum TellOpt')]
    decl = re.findall(r'X\(\w+, "(\w+)", (-?\d+)\)', shipped)
    assert [k for k, _ in decl] == keys
    assert {k: int(d) for k, d in decl} == hip.option_defaults()
    probe_keys = [k for k, _ in re.findall(r'X\(\w+, "(\w+)", (-?\d+)\)', probes)]
    assert {'q4_abl', 'pp2_abl', 'dcb_abl'} <= set(probe_keys)
    for k in probe_keys + ['no_such_option']:
        assert lib.tell_set_option(k.encode(), 1) == -1 and k.encode() in lib.tell_last_error()
        assert lib.tell_get_option(k.encode()) == -(1 << 63)
        with pytest.raises(KeyError):
            hip.set_option(k, 1)
    # round trip, context manager, the TELL_<KEY> spelling of the A/B scripts
    before = hip.get_option('gemm_s64')
    with hip.options(gemm_s64=2, conv_tile=3):
        assert hip.get_option('gemm_s64') == 2 and hip.get_option('conv_tile') == 3
    assert hip.get_option('gemm_s64') == before and hip.get_option('conv_tile') == hip.loaded_options()['conv_tile']
    hip.apply_env({'TELL_GEMM_RING': '4', 'TELL_STEP_GRAPH': '0'})          # (the second is a host-side switch: ignored)
    assert hip.get_option('gemm_ring') == 4
    hip.apply_env({'TELL_GEMM_RING': None})
    assert hip.get_option('gemm_ring') == hip.loaded_options()['gemm_ring']
    # a fresh process translates TELL_<KEY> once, at load time
    code = ('import sys; sys.path.insert(0, %r); import tell_amd; h = tell_amd.hip; h.lib(); '
            'print(h.get_option("gemm_q4"), h.get_option("q4_dynamic"))' % ROOT)
    env = dict(os.environ, TELL_GEMM_Q4='0', TELL_Q4_DYNAMIC='1', TELL_Q4_ABL='1')
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.split() == ['0', '1'], out.stdout + out.stderr
    # the binary: no TELL_* string (<= 10 allowed: error texts), no getenv among the imported symbols
    blob = open(hip.LIB_PATH, 'rb').read()
    assert blob.count(b'TELL_') <= 10
    nm = subprocess.run(['nm', '-D', '--undefined-only', hip.LIB_PATH], capture_output=True, text=True)
    if nm.returncode == 0:
        assert not [ln for ln in nm.stdout.splitlines() if re.search(r'\bgetenv\b', ln)], nm.stdout


def test_product_fails_loudly_without_gpu():
    import tell_amd
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError):
        tell_amd.hip.require_gpu()
    from tell_amd.models.resnet import ResNetFeatureExtractor
    m = ResNetFeatureExtractor((1, 1, 1, 1), width=8)
    with pytest.raises(RuntimeError):
        m(torch.randn(1, 3, 224, 224))             # no CPU fallback


def test_rng_restatement_matches_c():
    import tell_amd
    from tell_amd import rng
    lib = tell_amd.hip.lib()
    idx = np.array([0, 1, 2, 3, 4, 5, 6, 7, 63, 64, 1001, 1002, 2 ** 31 - 1, 2 ** 32 + 5, 2 ** 34 + 6, 2 ** 40 + 123,
                    2 ** 63 - 1], dtype=np.uint64)
    for seed, salt in [(0, 0), (0x5EED, 1), (123456789, 4000000000), (0xFFFFFFFF, 0xFFFFFFFF)]:
        want = np.array([lib.tell_keep_field_host(seed, salt, int(i)) for i in idx], dtype=np.uint64)
        got = rng.keep_field(seed, salt, idx)
        assert (got == want).all(), (seed, salt)
    for p in (0.0, 0.1, 0.25, 0.5, 0.999):
        assert int(rng.threshold(p)) == lib.tell_drop_threshold_host(p)


def test_rng_statistics():
    """The quad hash shares one mix between four decisions: keep rate, the joint distribution of a quad's four
    decisions, correlations along the index and across salts / seeds / graph steps have to look independent."""
    from tell_amd import rng
    n = 1 << 22
    sd = 1.0 / np.sqrt(n)
    for seed, salt in ((1, 2), (12345, 999), (0, 0)):
        for p in (0.1, 0.5):
            k = rng.keep_mask(seed, salt, n, p)
            kp = 1.0 - float(rng.threshold(p)) / 65536.0
            assert abs(k.mean() - kp) < 5 * np.sqrt(kp * (1 - kp)) * sd
            q = k.reshape(-1, 4).astype(np.int64)
            cnt = np.bincount(q[:, 0] + 2 * q[:, 1] + 4 * q[:, 2] + 8 * q[:, 3], minlength=16).astype(float)
            exp = np.array([np.prod([kp if (i >> j) & 1 else 1 - kp for j in range(4)]) for i in range(16)]) * len(q)
            assert ((cnt - exp) ** 2 / exp).sum() < 45.0           # chi2, 15 dof: p ~ 1e-4
            kc = k - k.mean()
            for lag in (1, 2, 3, 4, 8, 512, 513):
                assert abs(float((kc[:-lag] * kc[lag:]).mean() / kc.var())) < 6 * sd
    base = rng.keep_mask(5, 100, n, 0.1)
    step_salt = (100 + 1 * 0x632BE5AB) & 0xFFFFFFFF                # tell_step_salt: the next replay of a captured graph
    for other in (rng.keep_mask(5, 101, n, 0.1), rng.keep_mask(6, 100, n, 0.1), rng.keep_mask(5, step_salt, n, 0.1)):
        assert abs(np.corrcoef(base, other)[0, 1]) < 6 * sd


def test_warmup_linear_and_flat_layout():
    from tell_amd.training.optimizers import FlatParams, warmup_linear
    assert warmup_linear(0.0, 0.05) == 0.0                  # first BertAdam step has lr 0
    assert abs(warmup_linear(0.025, 0.05) - 0.5) < 1e-12
    assert abs(warmup_linear(0.05, 0.05) - 1.0) < 1e-12
    assert abs(warmup_linear(0.525, 0.05) - 0.5) < 1e-12
    assert warmup_linear(1.5, 0.05) == 0.0
    lin = torch.nn.Linear(5, 3)
    tied = torch.nn.Linear(5, 3)
    tied.weight = lin.weight
    frozen = torch.nn.Parameter(torch.ones(7), requires_grad=False)
    named = [('a.weight', lin.weight), ('a.bias', lin.bias), ('b.weight', tied.weight), ('f', frozen)]
    before = lin.weight.detach().clone()
    flat = FlatParams(named, 'cpu')
    assert flat.names == ['a.weight', 'a.bias']             # tied + frozen parameters appear once / never
    assert flat.total == 2048 and flat.n_chunks == 2        # CHUNK-aligned starts
    assert torch.equal(lin.weight.detach(), before)
    assert lin.weight.data_ptr() == flat.flat.data_ptr()
    assert lin.weight.grad.data_ptr() == flat.grad.data_ptr()
    assert flat.chunk_tensor.tolist() == [0, 1] and flat.chunk_begin.tolist() == [0, 1, 2]


def _write_cfg(tmp_path, kind):
    import yaml
    from tell_amd.build import decoder_kwargs
    dec = decoder_kwargs(vocab_size=600, dim=64, heads=4, ffn=128, cutoff=(100, 300))
    dec['type'] = 'dynamic_conv_decoder_' + kind
    dec['embedder'] = {
        'type': 'sum',
        'token_embedders': {
            'adaptive': {'type': 'adaptive', 'vocab_size': 600, 'namespace': 'bpe', 'initial_dim': 64,
                         'output_dim': 64, 'factor': 1, 'cutoff': [100, 300], 'padding_idx': 0, 'scale_embeds': True},
            'position': {'type': 'sinusoidal_positional', 'init_size': 512, 'embedding_dim': 64, 'padding_idx': 1,
                         'left_pad': False}},
        'embedder_to_indexer_map': {'adaptive': ['roberta'], 'position': ['roberta']}, 'allow_unmatched_keys': True}
    cfg = {'model': {'type': 'transformer_' + ('faces_objects' if kind == 'faces_objects' else 'flattened'),
                     'decoder': dec, 'criterion': {'type': 'adaptive_loss', 'padding_idx': 1},
                     'evaluate_mode': False, 'sampling_topk': 1, 'vocab_size': 600, 'weigh_bert': True,
                     'padding_value': 1, 'index': 'roberta', 'namespace': 'bpe'},
           'trainer': {'type': 'callback_apex', 'optimizer': {'type': 'bert_adam', 'lr': 1e-4},
                       'no_grad': ['^resnet', '^roberta']}}
    path = tmp_path / 'config.yaml'
    path.write_text(yaml.safe_dump(cfg))
    return str(path)


@pytest.mark.parametrize('kind', ['flattened', 'faces_objects'])
def test_from_params_shim(tmp_path, kind):
    from tell_amd import config
    from tell_amd.models import DynamicConvDecoder, DynamicConvFacesObjectsDecoder
    model, params = config.from_config(_write_cfg(tmp_path, kind), overrides='{"model": {"weigh_bert": false}}',
                                       resnet=object(), roberta=object())
    assert isinstance(model.decoder, DynamicConvFacesObjectsDecoder if kind == 'faces_objects' else DynamicConvDecoder)
    assert model.weigh_bert is False and params['trainer']['type'] == 'callback_apex'
    keys = set(model.state_dict())
    assert 'decoder.layers.3.context_attns.article.bias_k' in keys
    assert 'decoder.adaptive_softmax.tail.1.2.weight' in keys
    assert ('decoder.layers.0.context_attns.obj.k_proj_weight' in keys) == (kind == 'faces_objects')


REF_CFG = '/root/reference/expt/nytimes/9_transformer_objects/config.yaml'


@pytest.mark.skipif(not os.path.exists(REF_CFG), reason='reference tree only exists in the authoring container')
def test_reference_expt_configs_instantiate_unmodified():
    from tell_amd import config
    model, params = config.from_config(REF_CFG, resnet=object(), roberta=object())
    assert sum(p.numel() for p in model.decoder.parameters()) == 200461312
    assert model.weigh_bert and model.bert_weight.numel() == 25
    m2, _ = config.from_config(REF_CFG.replace('9_transformer_objects', '5_transformer_roberta'),
                               resnet=object(), roberta=object())
    assert m2.decoder.layers[0].context_names == ['image', 'article']
    # the context-table siblings and every other config of the transformer family
    m3, _ = config.from_config(REF_CFG.replace('9_transformer_objects', '8_transformer_faces'),
                               resnet=object(), roberta=object())
    assert type(m3).__name__ == 'TransformerFacesModel'
    assert m3.decoder.layers[0].context_names == ['image', 'article', 'faces']
    m4, _ = config.from_config(REF_CFG.replace('9_transformer_objects', '4_no_image'),
                               resnet=object(), roberta=object())
    assert m4.decoder.layers[0].context_names == ['article']
    for name in ('6_transformer_weighted_roberta', '7_transformer_location_aware'):
        mm, _ = config.from_config(REF_CFG.replace('9_transformer_objects', name), resnet=object(), roberta=object())
        assert mm.decoder.layers[0].context_names == ['image', 'article'], name
    for ds in ('goodnews',):
        for name in ('4_no_image', '5_transformer_roberta', '6_transformer_weighted_roberta', '8_transformer_faces',
                     '9_transformer_objects'):
            config.from_config(REF_CFG.replace('nytimes', ds).replace('9_transformer_objects', name),
                               resnet=object(), roberta=object())
    # the GloVe/LSTM baseline: `baseline_glove` + `lstm_decoder_flattened`
    for ds in ('nytimes', 'goodnews'):
        mb, _ = config.from_config(REF_CFG.replace('nytimes', ds).replace('9_transformer_objects', '1_lstm_glove'),
                                   resnet=object())
        assert type(mb).__name__ == 'BaselineGloveModel' and type(mb.decoder).__name__ == 'LSTMDecoder'
        assert len(mb.decoder.layers) == 4 and mb.decoder.hidden_size == 1536 and mb.max_caption_len == 50
        assert mb.decoder.article_attention.input_proj.weight_v.shape == (300, 1536)
        # 2_transformer_glove: the flattened decoder over GloVe vectors; 3_lstm_roberta: the LSTM decoder behind the
        # RoBERTa model class
        mg, _ = config.from_config(REF_CFG.replace('nytimes', ds).replace('9_transformer_objects', '2_transformer_glove'),
                                   resnet=object())
        assert type(mg).__name__ == 'TransformerGloveModel'
        assert mg.decoder.layers[0].context_attns['article'].kdim == 300
        ml, _ = config.from_config(REF_CFG.replace('nytimes', ds).replace('9_transformer_objects', '3_lstm_roberta'),
                                   resnet=object(), roberta=object())
        assert type(ml).__name__ == 'TransformerFlattenedModel' and type(ml.decoder).__name__ == 'LSTMDecoder'
    from tell_amd.common.registrable import Registrable
    from tell_amd.training.trainer import TrainerBase
    assert TrainerBase.by_name(params['trainer']['type']).__name__ == 'CallbackApexTrainer'
    assert isinstance(model, Registrable)


def test_signature_cache_lru_and_thrash_guard():
    """graphs.SignatureCache: least recently used READY entries are evicted down to the capacity; a rotation of more
    signatures than the capacity (every miss an eviction) freezes the set after THRASH_EVICTIONS evictions within
    THRASH_WINDOW sightings - make_room() then says "do not capture" - and thaws THRASH_FREEZE sightings later."""
    from tell_amd import graphs
    c = graphs.SignatureCache(4, capture_after=1)

    def sight(sig):
        e = c.touch(sig)
        if e['state'] == 'seen' and c.due(e):
            if c.make_room():
                e['state'] = 'ready'
                return 'captured'
            return 'eager'
        return 'replayed' if e['state'] == 'ready' else 'eager'
    assert [sight(i) for i in range(4)] == ['captured'] * 4 and c.evictions == 0
    assert [sight(i) for i in range(4)] == ['replayed'] * 4
    assert sight(10) == 'captured' and c.evictions == 1 and 0 not in c.entries          # LRU victim: signature 0
    assert sight(1) == 'replayed'
    # 12 signatures in rotation over 4 slots: every sighting misses
    seen = [sight(100 + (i % 12)) for i in range(40)]
    # (one eviction of the window is already spent above)
    assert c.freezes == 1 and seen.count('captured') == graphs.THRASH_EVICTIONS - 1 and 'eager' in seen[-10:]
    kept = [k for k, v in c.entries.items() if v['state'] == 'ready']
    assert len(kept) == 4
    ev = c.evictions
    assert [sight(k) for k in kept] == ['replayed'] * 4                                   # the frozen set keeps replaying
    i = 0
    while c.clock < c.frozen_until - 1:                                                   # ... others stay eager, nothing is evicted
        assert sight(1000 + i % 7) == 'eager'
        i += 1
    assert i > graphs.THRASH_FREEZE // 2 and c.evictions == ev
    assert sight(2000) == 'captured' and c.evictions == ev + 1                            # thawed: the LRU gets another chance
