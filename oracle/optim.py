"""BertAdam restated on CPU (pytorch_pretrained_bert.BertAdam, third-party, absent here;
**parity unpinned** - restated from the library's documented update rule; call site
expt/nytimes/9_transformer_objects/config.yaml:126-136).  TEST INFRASTRUCTURE."""
import torch


def warmup_linear(progress, warmup):
    if progress < warmup:
        return progress / warmup
    return max((progress - 1.0) / (warmup - 1.0), 0.0)


class BertAdam:
    def __init__(self, params, lr=1e-4, warmup=0.05, t_total=437600, b1=0.9, b2=0.98, e=1e-6,
                 weight_decay=1e-5, max_grad_norm=0.1):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.warmup, self.t_total = lr, warmup, t_total
        self.b1, self.b2, self.e, self.wd, self.max_grad_norm = b1, b2, e, weight_decay, max_grad_norm
        self.state = {id(p): dict(step=0, m=torch.zeros_like(p), v=torch.zeros_like(p)) for p in self.params}

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        for p in self.params:
            if p.grad is None:
                continue
            st = self.state[id(p)]
            g = p.grad
            if self.max_grad_norm > 0:                       # per-TENSOR clip_grad_norm_
                coef = self.max_grad_norm / (g.norm() + 1e-6)
                if coef < 1:
                    g = g * coef
            st['m'].mul_(self.b1).add_(g, alpha=1 - self.b1)
            st['v'].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            update = st['m'] / (st['v'].sqrt() + self.e)
            if self.wd > 0:
                update = update + self.wd * p
            lr = self.lr
            if self.t_total != -1:
                lr = lr * warmup_linear(st['step'] / self.t_total, self.warmup)
            p.add_(update, alpha=-lr)
            st['step'] += 1
