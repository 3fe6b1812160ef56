"""Architecture restatements of the two frozen, third-party encoders.

TEST INFRASTRUCTURE - see oracle/__init__.py.  **Parity unpinned**: neither
torchvision (0.6.1, environment.yml:19) nor fairseq (@2f7e3f3323 via torch.hub,
transformer_faces_objects.py:49-50) is present in this image and no pretrained
weights can be downloaded; the reference's own tests pin nothing here.  What IS
checked: ResNet-152 against the layer/shape table of tell/models/resnet.py:12-117
(+ torchvision Bottleneck v1.5, stride on the 3x3), and RoBERTa against
`transformers.RobertaModel` (same published architecture) in
tests/test_oracle_encoders.py.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- ResNet-152
class Bottleneck(nn.Module):
    """torchvision Bottleneck v1.5 (1x1 -> 3x3/stride -> 1x1 x4, BN after each, identity or
    1x1/stride downsample); state_dict names conv{1,2,3}, bn{1,2,3}, downsample.{0,1}."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + idt)


class ResNetFeatureExtractor(nn.Module):
    """tell/models/resnet.py:12-117: trunk without avgpool/fc in forward ([B,2048,7,7]);
    the unused `fc` stays in the state_dict (:48)."""

    def __init__(self, layers=(3, 8, 36, 3), width=64):
        super().__init__()
        self.inplanes = width
        self.conv1 = nn.Conv2d(3, width, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.layer1 = self._make(width, layers[0], 1)
        self.layer2 = self._make(width * 2, layers[1], 2)
        self.layer3 = self._make(width * 4, layers[2], 2)
        self.layer4 = self._make(width * 8, layers[3], 2)
        self.fc = nn.Linear(width * 8 * 4, 1000)
        for m in self.modules():                                      # resnet.py:50-56
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def _make(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * 4))
        seq = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        seq += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)     # resnet.py:94-99
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))  # :101-108


def resnet152():
    return ResNetFeatureExtractor((3, 8, 36, 3))


# ----------------------------------------------------------------------------- RoBERTa-large
class _SelfAttn(nn.Module):
    def __init__(self, E):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * E, E).normal_(0, 0.02))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * E))
        self.out_proj = nn.Linear(E, E)
        nn.init.normal_(self.out_proj.weight, 0, 0.02)
        nn.init.zeros_(self.out_proj.bias)


class _EncLayer(nn.Module):
    def __init__(self, E, FF):
        super().__init__()
        self.self_attn = _SelfAttn(E)
        self.self_attn_layer_norm = nn.LayerNorm(E)
        self.fc1 = nn.Linear(E, FF)
        self.fc2 = nn.Linear(FF, E)
        self.final_layer_norm = nn.LayerNorm(E)
        for l in (self.fc1, self.fc2):
            nn.init.normal_(l.weight, 0, 0.02)
            nn.init.zeros_(l.bias)


class _SentenceEncoder(nn.Module):
    def __init__(self, V, E, FF, L, max_pos, pad):
        super().__init__()
        self.embed_tokens = nn.Embedding(V, E, pad)
        self.embed_positions = nn.Embedding(max_pos + pad + 1, E, pad)
        nn.init.normal_(self.embed_tokens.weight, 0, 0.02)
        nn.init.normal_(self.embed_positions.weight, 0, 0.02)
        self.embed_tokens.weight.data[pad].zero_()
        self.embed_positions.weight.data[pad].zero_()
        self.emb_layer_norm = nn.LayerNorm(E)
        self.layers = nn.ModuleList([_EncLayer(E, FF) for _ in range(L)])


class RobertaEncoder(nn.Module):
    """fairseq `roberta.large` feature extractor as the model calls it:
    `extract_features(ids, return_all_hiddens=True)` -> embedding output + one tensor per
    layer, each [B,S,E] (transformer_faces_objects.py:352-353).  Post-LN BERT blocks,
    erf-GELU, learned positions offset by the padding index, q scaled by head_dim^-0.5,
    padded positions zeroed after the embedding LayerNorm.  Parameter names follow fairseq's
    `model.decoder.sentence_encoder.*`."""

    def __init__(self, vocab=50265, dim=1024, ffn=4096, layers=24, heads=16, max_positions=512, pad=1,
                 dropout=0.1, attention_dropout=0.1):
        super().__init__()
        self.dim, self.heads, self.pad = dim, heads, pad
        self.dropout, self.attention_dropout = dropout, attention_dropout
        self.model = nn.Module()
        self.model.decoder = nn.Module()
        self.model.decoder.sentence_encoder = _SentenceEncoder(vocab, dim, ffn, layers, max_positions, pad)

    def extract_features(self, ids, return_all_hiddens=False):
        enc = self.model.decoder.sentence_encoder
        E, H = self.dim, self.heads
        pad_mask = ids.eq(self.pad)
        nonpad = (~pad_mask).long()
        pos = torch.cumsum(nonpad, dim=1) * nonpad + self.pad
        x = enc.emb_layer_norm(enc.embed_tokens(ids) + enc.embed_positions(pos))
        x = F.dropout(x, self.dropout, self.training)
        x = x * (~pad_mask).unsqueeze(-1).type_as(x)
        hiddens = [x]
        B, S, _ = x.shape
        for layer in enc.layers:
            a = layer.self_attn
            qkv = F.linear(x, a.in_proj_weight, a.in_proj_bias)
            q, k, v = qkv.split(E, dim=-1)
            q = q * (E // H) ** -0.5
            sh = lambda t: t.view(B, S, H, E // H).transpose(1, 2)               # noqa: E731
            sc = torch.matmul(sh(q), sh(k).transpose(-1, -2))
            sc = sc.masked_fill(pad_mask[:, None, None, :], float('-inf'))
            pr = F.dropout(torch.softmax(sc.float(), -1).type_as(sc), self.attention_dropout, self.training)
            o = torch.matmul(pr, sh(v)).transpose(1, 2).reshape(B, S, E)
            o = F.dropout(a.out_proj(o), self.dropout, self.training)
            x = layer.self_attn_layer_norm(x + o)
            h = layer.fc2(F.gelu(layer.fc1(x)))
            x = layer.final_layer_norm(x + F.dropout(h, self.dropout, self.training))
            hiddens.append(x)
        return hiddens if return_all_hiddens else hiddens[-1]


def roberta_large(**kw):
    return RobertaEncoder(**kw)
