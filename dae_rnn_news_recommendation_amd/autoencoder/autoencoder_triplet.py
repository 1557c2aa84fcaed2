"""``DenoisingAutoencoderTriplet`` -- the paper-style DAE with EXPLICIT (anchor, positive, negative) inputs
(reference ``autoencoder/autoencoder_triplet.py:14-315``) on the MI355X kernels.

Three row blocks (org, pos, neg) go through the same tied weights; autoencoder loss = sum of the three
unweighted row means (:303-305); triplet loss = mean softplus(h.h_neg - h.h_pos) (:308-311); cost = AE + alpha *
triplet (:314).  On the device the three blocks are stacked into one [3B x F] batch: every GEMM of the step runs
once with M = 3B, the explicit-triplet kernel adds d loss/d h per block (``dae_explicit_triplet``).

The shipped reference class cannot train (it reads ``self.train_summary`` that is never set, :146, and
``gen_batches_triplet`` forgets the int() cast, utils.py:86-90 -- SURVEY appendix B); this class implements the
behaviour its code describes.
"""
from __future__ import annotations

import time

import numpy as np
from scipy import sparse

from . import utils
from .autoencoder import DenoisingAutoencoder
from .. import _lib as L

__all__ = ["DenoisingAutoencoderTriplet"]


class DenoisingAutoencoderTriplet(DenoisingAutoencoder):
    def __init__(self, algo_name='dae_triplet', model_name='dae_triplet', compress_factor=10, main_dir='dae_triplet/',
                 enc_act_func='tanh', dec_act_func='none', loss_func='mean_squared', num_epochs=10, batch_size=10,
                 xavier_init=1, opt='gradient_descent', learning_rate=0.01, momentum=0.5, corr_type='none',
                 corr_frac=0., verbose=True, verbose_step=5, seed=-1, alpha=1, **kw):
        super().__init__(algo_name=algo_name, model_name=model_name, compress_factor=compress_factor, main_dir=main_dir,
                         enc_act_func=enc_act_func, dec_act_func=dec_act_func, loss_func=loss_func, num_epochs=num_epochs,
                         batch_size=batch_size, xavier_init=xavier_init, opt=opt, learning_rate=learning_rate,
                         momentum=momentum, corr_type=corr_type, corr_frac=corr_frac, verbose=verbose,
                         verbose_step=verbose_step, seed=seed, alpha=alpha, triplet_strategy='none', **kw)

    def _strategy_key(self):
        return "explicit"

    @staticmethod
    def _stack(d):
        blocks = [d[k] for k in ('org', 'pos', 'neg')]
        if isinstance(blocks[0], np.ndarray):
            return np.concatenate(blocks, axis=0)
        return sparse.vstack([sparse.csr_matrix(b) for b in blocks]).tocsr()

    def fit(self, train_set, validation_set=None, restore_previous_model=False):
        """``train_set``: dict {'org','pos','neg'} of same-shape matrices (reference :40-56)."""
        assert isinstance(train_set, dict) and all(k in train_set for k in ('org', 'pos', 'neg'))
        shape = train_set['org'].shape
        assert train_set['pos'].shape == shape and train_set['neg'].shape == shape
        N, n_features = shape
        self.sparse_input = not isinstance(train_set['org'], np.ndarray)
        self.n_components = int(np.floor(n_features / self.compress_factor))
        batch = self._resolve_batch(N)
        stacked = self._stack(train_set)                    # rows [0,N) org, [N,2N) pos, [2N,3N) neg
        eng = self._build_engine(n_features, 3 * batch, data=stacked)
        eng.upload_csr(stacked) if self.sparse_input else eng.upload_dense(stacked)
        eng.set_params(*self._initial_parameters(n_features))
        if restore_previous_model:
            self._restore(self.model_path)
        self._write_parameter_to_file(restore_previous_model)
        if validation_set is not None:
            assert isinstance(validation_set, dict) and all(k in validation_set for k in ('org', 'pos', 'neg'))
            assert validation_set['pos'].shape == validation_set['org'].shape == validation_set['neg'].shape
            assert validation_set['org'].shape[1] == n_features
        self._val_engine = None
        self._train_triplet(stacked, N, batch, validation_set)
        self._save(self.model_path)

    def _train_triplet(self, stacked, N, batch, validation_set):
        import torch
        eng = self.engine
        n_batches = -(-N // batch)
        self._stats = torch.zeros((max(self.num_epochs, 1), n_batches, L.STATS_STRIDE), dtype=torch.float32, device=eng.device)
        from .autoencoder import _EpochFeeder, pinned_copy, upload_ahead, uploaded

        def draw_epoch(e):
            # feeder thread, one epoch ahead: ONE corruption draw over the stacked set (org, pos, neg in order), then ONE shuffle
            # shared by the three blocks (utils.py:87-91); the batch row lists [idx | N + idx | 2N + idx] staged as one pinned array
            d = self._draw_epoch(stacked, e, n_shuffle=N)
            o = d['order'].astype(np.int32)
            rows = [np.concatenate([o[s:s + batch], N + o[s:s + batch], 2 * N + o[s:s + batch]]) for s in range(0, N, batch)]
            d['row_offsets'] = np.cumsum([0] + [len(r) for r in rows])
            d['rows_t'] = pinned_copy(np.concatenate(rows))
            if 'bits' in d:
                d['bits_t'] = pinned_copy(d['bits'])
            return upload_ahead(d, dev, ('rows_t', 'bits_t'))       # ... and on their way to the device, on a copy stream, an epoch early

        dev = eng.device
        feeder = _EpochFeeder(draw_epoch, self.num_epochs)
        t_fit = time.time()
        t_first = None
        try:
            for i in range(self.num_epochs):
                t0 = time.time()
                draw = feeder.get()
                plan = self._corruption_plan(draw, i)
                sp = plan.pop('_sp', None)                   # device salt-and-pepper (rng='philox'): flipped per batch
                if sp is not None and not hasattr(self, '_sp_range'):
                    d = stacked.data if stacked.nnz else np.zeros(1)
                    full = stacked.nnz == stacked.shape[0] * stacked.shape[1]
                    self._sp_range = (float(d.min() if full else min(d.min(), 0.0)), float(d.max() if full else max(d.max(), 0.0)))
                rows_dev = uploaded(draw, 'rows_t', eng.device)
                off = draw['row_offsets']
                for b in range(n_batches):
                    rows = rows_dev[off[b]:off[b + 1]]
                    if sp is not None:
                        plan['corrupted_csr'] = eng.salt_pepper_batch(rows, sp[0], self._sp_range[0], self._sp_range[1], sp[1], sp[2])
                    eng.train_step(rows, None, self._stats[i, b], phase=0, **plan)
                if (i + 1) % self.verbose_step == 0 or i + 1 == self.num_epochs:
                    torch.cuda.synchronize()
                    self.train_time = time.time() - t0
                    self._run_validation_error_and_summaries(i + 1, validation_set, None)
                if i == 0:
                    torch.cuda.synchronize()
                    t_first = time.time()
        finally:
            feeder.close()
        torch.cuda.synchronize()
        t_end = time.time()
        if self.num_epochs > 1 and t_end > t_first:          # rows x timed epochs / wall, first epoch excluded as warm-up
            self.samples_per_sec = 3 * N * (self.num_epochs - 1) / (t_end - t_first)
        elif self.num_epochs > 0 and t_end > t_fit:
            self.samples_per_sec = 3 * N * self.num_epochs / (t_end - t_fit)

    def _validation_forward_triplet(self, validation_set):
        """Forward pass of the whole validation dict as ONE stacked batch, uncorrupted (reference :160-199)."""
        import torch
        from ..engine import Engine
        stacked = self._stack(validation_set)
        nv = validation_set['org'].shape[0]
        if getattr(self, '_val_engine', None) is None:
            act = lambda a: a if a in ('sigmoid', 'tanh') else 'none'
            self._val_engine = Engine(stacked.shape[1], self.n_components, 3 * nv, dtype=self._forward_precision(stacked), enc_act=act(self.enc_act_func),
                                      dec_act=act(self.dec_act_func), loss_func=self.loss_func, opt='gradient_descent',
                                      alpha=float(self.alpha), triplet='explicit', device=self.device)
            self._val_engine.upload_dense(stacked) if isinstance(stacked, np.ndarray) else self._val_engine.upload_csr(stacked)
        ve = self._val_engine
        ve.set_params(*self.engine.get_params())
        stats = torch.zeros(L.STATS_STRIDE, dtype=torch.float32, device=ve.device)
        ve.train_step(torch.arange(3 * nv, dtype=torch.int32, device=ve.device), None, stats, phase=2)
        return stats.cpu().numpy()

    def _run_validation_error_and_summaries(self, epoch, validation_set, validation_set_label):
        st = self.epoch_stats(epoch)
        rec = dict(epoch=epoch, seconds=self.train_time, cost=st['cost'], ae=st['ae'], triplet=st['triplet'])
        if self.verbose == 1:
            print('At step %d (%.2f seconds): ' % (epoch, self.train_time), end='')
            print('[Train Stat (average over past steps)] - Cost: ', end='')
            print('Overall=%.4f\tAutoencoder=%.4f\tTriplet=%.4f\t' % (st['cost'], st['ae'], st['triplet']), end='')
        if validation_set is None:
            if self.verbose == 1:
                print()
            self.history.append(rec)
            return
        v = self._validation_forward_triplet(validation_set)          # reference :186-199
        rec.update(val_cost=float(v[L.STAT_COST]), val_ae=float(v[L.STAT_AE]), val_triplet=float(v[L.STAT_TRIPLET]))
        if self.verbose:
            print("[Validation Stat (at this step)] - Cost: ", end='')
            print('Overall=%.4f\tAutoencoder=%.4f\tTriplet=%.4f\t' % (v[L.STAT_COST], v[L.STAT_AE], v[L.STAT_TRIPLET]))
        self.history.append(rec)
