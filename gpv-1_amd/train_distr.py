"""Training driver: the step loop around the hot path.

Reference: exp/gpv/train_distr.py (``train_worker`` :150-474, ``main`` :478-495) launched by scripts/train.sh.
What is reproduced -- everything between the data loader and the checkpoint file:
  * Hydra-style invocation: ``python -m gpv1_amd.train_distr [--config some.yaml] key=value ...``
    (config.py; the reference's own YAML loads too);
  * model construction ``GPV(cfg.model)``, optional ``load_pretr_detr()`` (:182-183), phase-1 freeze of the DETR
    parameters that came from the checkpoint (``freeze_detr_params`` :136-140, ``training.freeze`` -> ``frozen_epochs`` /
    ``frozen_batch_size`` :318-320,482-484);
  * the four AdamW groups, clip at ``clip_max_norm`` on DETR parameters, warm-up-linear schedule stepped every iteration
    (:228-253,298-313,421-428,468-469) -- all inside FlatTrainer;
  * one process per GPU (RANK / WORLD_SIZE / LOCAL_RANK from the launcher; the reference ``mp.spawn``s, :493), per-rank
    batch = ``batch_size // world`` (:490), DistributedSampler-style disjoint index shards reshuffled per epoch (:202,396-397);
  * checkpoints in the reference layout (:381-389): ``model`` with ``module.``-prefixed keys (what a DDP-wrapped model
    saves and ``inference.py:59-60`` strips), ``optimizer``, ``epoch``, ``step``, ``lr``, ``model_selection_metric``,
    ``warmup_scheduler``; resume takes every key whose name and size match (:264-271).
Out of scope: dataset ETL, evaluators (the reference checkpoints only when its validation metric improves;
without evaluators this driver checkpoints every ``training.ckpt_step`` steps and at every epoch end), TensorBoard, HTML.
The dataset is any sequence of ``(image[3,H,W] fp32 normalised, query str | (ids, mask), target dict)``;
``SyntheticCocoDataset`` provides BASELINE's synthetic COCO-shaped samples.
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

from .config import from_dict, load_config
from .default_config import default_tree, GROUP_OPTIONS
from .gpv import GPV
from .misc import nested_tensor_from_tensor_list
from .train import FlatTrainer


def freeze_detr_params(model, requires_grad=False):
    """train_distr.py:136-140"""
    for n, p in model.named_parameters():
        if n in model.init_detr_params:
            p.requires_grad = requires_grad


class SyntheticCocoDataset:
    """SURVEY §8(d) synthetic samples: N(0,1) images, random query ids, the four task target types round-robin."""

    def __init__(self, n, vocab, image_size=(480, 640), query_len=6, seed=0, tasks=('CocoCaptioning', 'CocoVqa', 'CocoClassification', 'CocoDetection')):
        self.n, self.vocab, self.size, self.tl, self.seed, self.tasks = n, vocab, image_size, query_len, seed, tasks

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        img = torch.randn(3, *self.size, generator=g)
        ids = torch.randint(1000, 30000, (self.tl,), generator=g)
        task = self.tasks[i % len(self.tasks)]
        words = [self.vocab[int(j)] for j in torch.randint(0, len(self.vocab) - 4, (19 if task == 'CocoCaptioning' else 2,), generator=g)]
        t = {'task': task}
        if task == 'CocoDetection':
            nb = int(torch.randint(1, 11, (1,), generator=g))
            cxcy = 0.25 + 0.5 * torch.rand(nb, 2, generator=g)
            wh = 0.05 + 0.3 * torch.rand(nb, 2, generator=g)
            t.update(boxes=torch.cat([cxcy, wh], 1), labels=torch.zeros(nb, dtype=torch.long))
        else:
            t['answer'] = ' '.join(words)
        return img, (ids, torch.ones(self.tl, dtype=torch.long)), t


def shard_indices(n, epoch, rank, world, seed=0):
    """torch.utils.data.DistributedSampler(shuffle=True): permutation seeded by (seed + epoch), padded to a multiple of
    world by wrapping, rank takes every world-th element."""
    g = torch.Generator().manual_seed(seed + epoch)
    idx = torch.randperm(n, generator=g).tolist()
    total = -(-n // world) * world
    idx += idx[:total - n]
    return idx[rank:total:world]


def batches(dataset, indices, batch_size, device):
    for s in range(0, len(indices) - batch_size + 1, batch_size):
        items = [dataset[i] for i in indices[s:s + batch_size]]
        imgs = [it[0].to(device) for it in items]
        qs = [it[1] for it in items]
        if isinstance(qs[0], str):
            queries = qs
        else:
            queries = (torch.stack([q[0] for q in qs]).to(device), torch.stack([q[1] for q in qs]).to(device))
        targets = []
        for it in items:
            targets.append({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in it[2].items()})
        yield nested_tensor_from_tensor_list(imgs), queries, targets


def save_checkpoint(path, model, trainer, epoch, step, metric=0.0):
    """train_distr.py:381-389 (DDP state dict => 'module.' prefix)"""
    sd = {'module.' + k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    tmp = path + '.tmp'
    torch.save({'model': sd, 'optimizer': trainer.state_dict(), 'epoch': epoch, 'step': step,
                'lr': list(trainer.current_lrs().values()), 'model_selection_metric': metric,
                'warmup_scheduler': {'last_epoch': trainer.step_count, 'warmup_steps': trainer.warmup_steps, 't_total': trainer.t_total}}, tmp)
    os.replace(tmp, path)


def load_checkpoint(path, model, trainer=None, map_location='cpu'):
    """train_distr.py:262-285 -- keys with or without the 'module.' prefix, only name+size matches are taken"""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    cur = model.state_dict()
    taken = 0
    for k, v in ckpt['model'].items():
        k = k[len('module.'):] if k.startswith('module.') else k
        if k in cur and cur[k].size() == v.size():
            cur[k] = v
            taken += 1
    model.load_state_dict(cur)
    if trainer is not None and ckpt.get('optimizer') is not None:
        trainer.load_state_dict(ckpt['optimizer'], step=ckpt.get('step'))      # torch.optim.AdamW layout (reference checkpoints) or round-1 layout
    return ckpt, taken


def train_worker(cfg, dataset=None, device=None, log=print):
    """one rank of the reference's ``train_worker``; returns (model, trainer, step)"""
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if device is None:
        device = f'cuda:{local}'
    if str(device).startswith('cuda'):
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '10001')                      # dist_url tcp://localhost:10001
        from .train import init_process_group
        init_process_group(rank, world, device)
    tr_cfg = cfg.training
    batch_size = tr_cfg.frozen_batch_size if tr_cfg.freeze else tr_cfg.batch_size
    per_rank = max(1, batch_size // world)
    epochs = tr_cfg.frozen_epochs if tr_cfg.freeze else tr_cfg.num_epochs

    model = GPV(cfg.model)
    if cfg.model.pretr_detr is not None and os.path.exists(str(cfg.model.pretr_detr)):
        model.load_pretr_detr()
    if tr_cfg.freeze:
        freeze_detr_params(model)
    model.to(device)
    if dataset is None:
        # the task mix: configs/learning_datasets/<name>.yaml selected by `learning_datasets=<name>` (scripts/train.sh:14-34),
        # CocoMultitaskDataset(cfg.learning_datasets, ...) in the reference (train_distr.py:163-166)
        tasks = tuple(cfg.get('learning_datasets', {}) or ()) or ('CocoCaptioning', 'CocoVqa', 'CocoClassification', 'CocoDetection')
        dataset = SyntheticCocoDataset(int(cfg.get('synthetic_samples', 4 * batch_size)), model.vocab, tasks=tasks)
    steps_per_epoch = (-(-len(dataset) // world)) // per_rank
    t_total = steps_per_epoch * epochs if tr_cfg.lr_linear_decay else 0
    have_ckpt = tr_cfg.ckpt is not None and os.path.exists(str(tr_cfg.ckpt))
    if not model.bert.pretrained and not have_ckpt:
        # the reference's query encoder is BertModel.from_pretrained('bert-base-uncased') (bert.py:8-9), frozen: training against
        # a random-init BERT is only meaningful for throughput / plumbing runs
        log(f'[rank {rank}] WARNING: BERT has random weights (no model.bert_weights / GPV_BERT_WEIGHTS and no checkpoint): '
            f'the frozen query encoder is NOT bert-base-uncased')
        if cfg.get('require_pretrained_bert', False):
            raise RuntimeError('training.require_pretrained_bert is set and no BERT weights were given')
    sched = {}
    if not tr_cfg.lr_linear_decay:                         # MultiStepLR per epoch x GradualWarmup over epoch 0 (train_distr.py:288-308)
        sched = {'lr_milestones': list(tr_cfg.get('lr_milestones', []) or []), 'lr_drop': tr_cfg.get('lr_drop', 0.1),
                 'warmup_iters': steps_per_epoch if tr_cfg.lr_warmup else 0}
    trainer = FlatTrainer(model, lr=tr_cfg.lr, lr_backbone=tr_cfg.lr_backbone, weight_decay=tr_cfg.weight_decay,
                          clip_max_norm=tr_cfg.clip_max_norm,
                          warmup_steps=int(tr_cfg.lr_warmup_fraction * t_total) if tr_cfg.lr_warmup else 0, t_total=t_total, **sched)
    step, last_epoch = 0, -1
    if have_ckpt:
        ckpt, taken = load_checkpoint(tr_cfg.ckpt, model, trainer, map_location=device)
        step, last_epoch = ckpt['step'], ckpt['epoch']
        log(f'[rank {rank}] resumed {tr_cfg.ckpt}: {taken} tensors, end of epoch {last_epoch}, step {step}')
    ckpt_path = os.path.join(cfg.ckpt_dir, 'model.pth')
    if rank == 0:
        os.makedirs(cfg.ckpt_dir, exist_ok=True)
    max_steps = cfg.get('max_steps', None)
    t0 = time.time()
    for epoch in range(last_epoch + 1, epochs):
        idx = shard_indices(len(dataset), epoch, rank, world)
        epoch_done = True                     # False: max_steps ended the epoch before its last batch
        batch_iter = iter(batches(dataset, idx, per_rank, device))
        for it, (imgs, queries, targets) in enumerate(batch_iter):
            trainer.set_epoch(epoch, it)
            loss = trainer.train_step(imgs, queries, targets)
            step += 1
            if rank == 0 and step % tr_cfg.log_step == 0:
                log(f'epoch {epoch} step {step} loss {float(loss.detach()) if loss is not None else float("nan"):.4f} '
                    f'lr {trainer.current_lrs()["others"]:.3e} {time.time() - t0:.1f}s')
            if rank == 0 and step % tr_cfg.ckpt_step == 0:
                # like the reference's mid-epoch save (train_distr.py:372-389: 'epoch': epoch-1 next to the CURRENT step): a
                # resume re-runs this epoch from its start while the schedule continues from `step` -- the reference's behaviour,
                # kept as is (with lr_linear_decay the tail of such a run sits at lr 0 once step passes t_total)
                save_checkpoint(ckpt_path, model, trainer, epoch - 1, step)
            if max_steps is not None and step >= max_steps:
                epoch_done = next(batch_iter, None) is None      # stopped on the epoch's last batch: the epoch is complete
                break
        stopped = max_steps is not None and step >= max_steps
        if rank == 0:
            save_checkpoint(ckpt_path, model, trainer, epoch if epoch_done else epoch - 1, step)
        if stopped:
            break
    if world > 1:
        dist.barrier()
        from .misc import note_sync_collective
        note_sync_collective()
    return model, trainer, step


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--config', default=None, help='YAML file (e.g. the reference configs/exp/gpv.yaml); default: gpv1_amd.default_config')
    ap.add_argument('overrides', nargs='*', help='Hydra-style key=value overrides')
    args = ap.parse_args(argv)
    cfg = load_config(args.config, args.overrides) if args.config else from_dict(default_tree(), args.overrides, group_options=GROUP_OPTIONS)
    train_worker(cfg)


if __name__ == '__main__':
    main(sys.argv[1:])
