"""Train / evaluate / test loops with the reference's behaviour
(/root/reference/MIND_2020/train_eval.py:35-153, 219-273, 300-341) on the HIP path.

train():    Adam(lr=config.learning_rate) + CrossEntropy with label 0, loss averaged every 100
            iterations, dev AUC every ``config.eval_step`` batches and at each epoch end, checkpoint
            ``T{time}_{model}_epoch{E}_iter_{n}_auc_{auc:.3f}.ckpt`` when the AUC improves
            (train_eval.py:139-149).  The step itself is ``model.train_step`` (fused HIP forward + CE +
            backward + Adam); ``use_autograd=True`` runs the reference's literal sequence instead
            (``model(batch)`` -> criterion -> backward -> torch.optim.Adam) through the same kernels.
evaluate(): per-impression AUC on the un-padded prefix, unweighted mean (train_eval.py:219-271), with
            the scores never leaving the GPU (``nrms_impression_auc``) instead of a fork pool.
test():     per-impression rank lists in the MIND submission format (train_eval.py:280-286,335-341).
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch
import torch.nn as nn

from . import parallel


def _inner(model):
    return model.model if hasattr(model, "model") and hasattr(model.model, "train_step") else model


def _pad_labels(y_true, max_c, device):
    n = len(y_true)
    lab = np.zeros((n, max_c), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.int32)
    for i, y in enumerate(y_true):
        k = min(len(y), max_c)
        lab[i, :k] = np.asarray(y[:k], dtype=np.uint8)
        lens[i] = k
    return torch.from_numpy(lab).to(device), torch.from_numpy(lens).to(device)


def evaluate(config, model, data_iter, y_true=None, AUC_best=None, verbose=True):
    """y_true: list (one entry per impression, in data_iter order) of 0/1 label lists -- the
    reference keeps it in the module global ``_y_true`` read from dev_behaviors.csv (:36-39); None reads
    that file (data_handler.read_dev_labels).

    Deviation, on purpose: the reference calls ``model.eval()`` here and never ``model.train()`` again (its
    re-enable is commented out, train_eval.py:122,230), so after its first evaluation it trains with dropout
    off.  This evaluate() restores the mode it found."""
    if y_true is None:
        from .data_handler import read_dev_labels
        y_true = read_dev_labels(config)
    net = _inner(model)
    was_training = net.training
    net.eval()
    scores = []
    eng = net.engine
    eng.news_cache_begin()                     # weights are constant from here to the end of the evaluation
    try:
        with torch.no_grad():
            for datas in data_iter:
                scores.append(net(datas))
    finally:
        net.last_eval_cache = eng.news_cache_end()
    rank_score = torch.cat(scores, dim=0)
    eng.check_ids()
    lab, lens = _pad_labels(y_true[:rank_score.shape[0]], rank_score.shape[1], rank_score.device)
    aucs = eng.impression_auc(rank_score, lab, lens)
    AUC = float(aucs.mean().item())
    net.last_eval_scores, net.last_eval_aucs = rank_score, aucs      # diagnostics (per-impression values of this evaluation)
    if verbose:
        print('AUC:', AUC)
    net.train(was_training)
    return AUC


def log_res(config, step, auc):
    if parallel.env_world()[0] != 0:             # one writer: every rank computes the same dev AUC
        return
    os.makedirs(config.log_path, exist_ok=True)
    with open(os.path.join(config.log_path, 'res.txt'), 'a+') as f:
        f.write('{}_{}_:auc_{}\n'.format(time.strftime('%m-%d_%H.%M'), auc, step))


def _save(config, model, total_batch, auc):
    os.makedirs(config.save_path, exist_ok=True)
    name = 'T{}_{}_epoch{}_iter_{}_auc_{:.3f}.ckpt'.format(time.strftime('%m-%d_%H.%M'), config.model_name,
                                                          config.num_epochs, total_batch, auc)
    torch.save(model.state_dict(), os.path.join(config.save_path, name))
    return name


def warmup_lr(base_lr, i, warm_up_steps):
    """Learning rate the reference's warm-up phase applies to iteration ``i`` (0-based): its
    GradualWarmupScheduler(multiplier=1, total_epoch=warm_up_steps) starts at 0 and is stepped with
    ``scheduler.step(i)`` AFTER iteration i (train_eval.py:66-71,99; lr_scheduler.py:41-42), so iteration
    i runs at base_lr * max(i - 1, 0) / warm_up_steps, capped at base_lr."""
    return base_lr * min(max(i - 1, 0) / float(warm_up_steps), 1.0)


def warmup_iterations(warm_up_steps=500):
    """The reference leaves the warm-up loop with ``if i > 500: break`` after running iteration i
    (train_eval.py:97-98): iterations 0..501 whatever warm_up_steps is."""
    return 502


def train(config, model, train_iter, dev_iter=None, dev_labels=None, use_autograd=False, max_batches=None,
          verbose=True):
    """Returns dict(losses=[...per batch...], aucs=[(batch, auc), ...], ckpts=[...]).
    With config.warm_up the epochs are preceded by the reference's warm-up pass over the first
    batches of train_iter with a linearly increasing learning rate (train_eval.py:64-99)."""
    net = _inner(model)
    rank, _, world = parallel.env_world()
    reduce = parallel.GradAllReduce() if world > 1 and torch.distributed.is_initialized() else None
    start = time.time()
    model.train()
    optimizer = criterion = None
    if use_autograd:
        optimizer = torch.optim.Adam(model.parameters(), lr=config.learning_rate)
        criterion = nn.CrossEntropyLoss()
        # this loop zeroes the gradients before every backward and keeps no reference to an old .grad: the backward may
        # hand autograd views of ONE persistent flat buffer instead of 57.6 MB of fresh memory per step (opt-in contract,
        # model/nrms_hip.py _NRMSFunction.backward)
        net.reuse_grad_buffer = True
    total_batch, AUC_best, STEP_SIZE = 0, 0.56, 100          # train_eval.py:59,61
    hist = dict(losses=[], aucs=[], ckpts=[], warmup_losses=[])
    window = []
    done = False
    if getattr(config, 'warm_up', False):
        if verbose:
            print('warm-up training...')
        wlosses = []
        for i, datas in enumerate(train_iter):
            lr_i = warmup_lr(config.learning_rate, i, config.warm_up_steps)
            if use_autograd:
                for grp in optimizer.param_groups:
                    grp['lr'] = lr_i
                outputs = model(datas)
                model.zero_grad()
                loss = criterion(outputs, torch.zeros(len(outputs), dtype=torch.long, device=outputs.device))
                loss.backward()
                optimizer.step()
                wlosses.append(loss.detach())
            else:
                wlosses.append(net.train_step(datas, lr=lr_i, world_size=world, all_reduce=reduce) /
                               len(datas['browsed_titles']))
            if i % 100 == 0 and verbose:
                print('Warm-up Steps: {0:>6},  Train Loss: {1:>5.6}'.format(i, float(wlosses[-1])))
            if i + 1 >= warmup_iterations(config.warm_up_steps) or (max_batches is not None and i + 1 >= max_batches):
                break
        hist['warmup_losses'] = [float(v) for v in wlosses]
        if use_autograd:
            for grp in optimizer.param_groups:
                grp['lr'] = config.learning_rate
    for epoch in range(config.num_epochs):
        if verbose:
            print('Epoch [{}/{}]'.format(epoch + 1, config.num_epochs))
        for datas in train_iter:
            B = len(datas['browsed_titles'])
            if use_autograd:
                outputs = model(datas)
                model.zero_grad()
                y = torch.zeros(len(outputs), dtype=torch.long, device=outputs.device)
                loss = criterion(outputs, y)
                loss.backward()
                optimizer.step()
                window.append(loss.detach())
            else:
                loss_sum = net.train_step(datas, world_size=world, all_reduce=reduce)
                window.append(loss_sum / B)
            if total_batch % STEP_SIZE == 0:                 # one host sync per 100 iterations, not two per step
                vals = [float(v) for v in window]
                # out-of-range word ids surface here: after the very first batch (total_batch == 0: a vocabulary / table
                # mismatch is caught before a second update is applied, where nn.Embedding would have raised) and then
                # every 100 steps
                net.engine.check_ids()
                hist['losses'].extend(vals)
                window = []
                if verbose:
                    print('Iter: {0:>6},  Train Loss: {1:>5.6},  Time: {2:.1f}s'.format(
                        total_batch, float(np.mean(vals)), time.time() - start))
            total_batch += 1
            if dev_iter is not None and total_batch % config.eval_step == 0:
                auc = evaluate(config, model, dev_iter, dev_labels, AUC_best, verbose)
                hist['aucs'].append((total_batch, auc))
                log_res(config, auc, total_batch)
                if auc > AUC_best:
                    AUC_best = auc
                    if config.save_flag and rank == 0:
                        hist['ckpts'].append(_save(config, model, total_batch, AUC_best))
            if max_batches is not None and total_batch >= max_batches:
                done = True
                break
        if dev_iter is not None:
            auc = evaluate(config, model, dev_iter, dev_labels, AUC_best, verbose)
            hist['aucs'].append((total_batch, auc))
            log_res(config, auc, 'epoch_{}'.format(epoch))
            if auc > AUC_best:
                AUC_best = auc
                if config.save_flag and rank == 0:
                    hist['ckpts'].append(_save(config, model, total_batch, AUC_best))
        if done:
            break
    hist['losses'].extend(float(v) for v in window)
    return hist


def _cal_test(scores, n):
    """Rank (1 = best) of each shown candidate (train_eval.py:280-286)."""
    res = np.argsort(-np.asarray(scores[:n]), kind="stable")
    rank = [0] * n
    for pos, v in enumerate(res):
        rank[v] = pos + 1
    return rank


def best_checkpoint(config):
    """The checkpoint of this model name with the highest dev AUC in its file name
    (``..._auc_0.673.ckpt``, written by train(); selection rule of train_eval.py:303-308, parsed as a float
    instead of ``eval``).  None if there is none above 0.5."""
    best, best_auc = None, 0.5
    if not os.path.isdir(config.save_path):
        return None
    for ckpt in sorted(os.listdir(config.save_path)):
        if config.model_name not in ckpt or not ckpt.endswith('.ckpt'):
            continue
        try:
            auc = float(ckpt[:-len('.ckpt')].split('_')[-1])
        except ValueError:
            continue
        if auc > best_auc:
            best, best_auc = ckpt, auc
    return best


def test(config, model, data_iter, test_list_nums=None, ckpt_file=None, out_file=None, pick_best=False):
    """Writes ``<impression index> [r1,r2,...]`` lines; returns the file name.  test_list_nums: shown
    candidates per impression (None: data_handler.get_Test_List, train_eval.py:287-298,315); pick_best: load the
    best checkpoint by file-name AUC when ckpt_file is None (train_eval.py:301-310)."""
    net = _inner(model)
    if ckpt_file is None and pick_best:
        ckpt_file = best_checkpoint(config)
        if ckpt_file is None:
            # the reference fails here too (it loads './save_model/' + None, train_eval.py:309): scoring whatever weights
            # the model happens to hold and writing a submission from them would be a silent wrong answer
            raise FileNotFoundError("test(pick_best=True): no checkpoint of model %r with a dev AUC above 0.5 in %r"
                                    % (config.model_name, config.save_path))
    if ckpt_file is not None:
        print('loading checkpoint:', os.path.join(config.save_path, ckpt_file))
        model.load_state_dict(torch.load(os.path.join(config.save_path, ckpt_file), weights_only=True))
    if test_list_nums is None:
        from .data_handler import get_Test_List
        test_list_nums = get_Test_List(config)
    net.eval()
    scores = []
    net.engine.news_cache_begin()
    try:
        with torch.no_grad():
            for datas in data_iter:
                scores.append(net(datas).cpu())
    finally:
        net.engine.news_cache_end()
    net.engine.check_ids()
    test_rank_score = np.concatenate([s.numpy() for s in scores])
    ranks = [_cal_test(test_rank_score[i], int(n)) for i, n in enumerate(test_list_nums[:len(test_rank_score)])]
    file_name = out_file or 'sumbit_{}_{}.txt'.format(config.model_name, time.strftime('%m-%d_%H.%M', time.localtime()))
    with open(file_name, 'w') as f:
        for i, r in enumerate(ranks):
            f.write(str(i + 1) + ' ')
            f.write(str(r).replace(' ', '') + '\n')
    return file_name
