"""``DenoisingAutoencoder`` -- drop-in for the reference estimator (``autoencoder/autoencoder.py:14``)
with the training step running as gfx950 HIP kernels (libdae_hip) instead of a TF1 graph.

Kept from the reference: constructor keywords and defaults (:20-23), ``fit`` / ``transform`` /
``load_model`` / ``get_model_parameters`` signatures, the ``results/<algo>/<main_dir>/{models,data,logs,
data/tsv,data/plot}/`` layout and the attributes the CLI reads, ``parameter.txt``, the per-epoch stdout
line (:283-294), the per-epoch order of host RNG draws (corrupt the whole set, then shuffle; :218-220)
and variable names of the checkpoint (``enc-w``, ``hidden-bias``, ``visible-bias``; :365-367).

New (keyword-only, all optional):
  precision   'auto' (default: per triplet strategy the cheapest mode measured to hold the reference's loss curve over 100 steps, _lib.AUTO_BY_STRATEGY:
              none -> 'f16x2d', batch_all / batch_hard -> 'f16x2h') or one of _lib.PRECISIONS: 'f16x2h' / 'f16x2d' (fp16 operand images; W
              and the h resp. delta2 operands as hi + lo), 'bf16x3' (split-bf16: every stored operand as hi + lo bf16, three products each), 'fp32'
              (exact-fp32 MFMA), 'f16x3' (every operand hi + lo fp16); 'f16x2' (W alone hi + lo: holds 20 steps, not 100), 'bf16' / 'f16' (single
              16-bit images, fp32 accumulate / master weights: faster, outside the gate)
  rng         'numpy'  -- reference-exact legacy-RandomState stream: keep decisions are drawn on the host
                          and shipped as one bit per stored entry per epoch;
              'philox' -- counter-based masking generated on the device (statistically equivalent,
                          no host RNG in the epoch loop)
  init_weights  (W0[, bh0, bv0]) injected instead of the Xavier draw (tf.random_uniform is not reproducible)
  data_parallel  True -> shard every mini-batch over torch.distributed ranks; dp_exchange 'auto': in the split-bf16 mode ONE fp32 all-reduce of
                 the flat gradient + the full optimizer step on every rank (dp.AllReduceExchange), in the bf16 / fp32 modes reduce-scatter of the
                 W gradient, sharded optimizer, all-gather of the low-precision shadow (dp.ShardedExchange); 'sharded' / 'allreduce' force one
  plan_options   {name: value} handed to dae_plan_set_option (implementation choices of the same arithmetic; A/B runs)
"""
from __future__ import annotations

import os
import time

import numpy as np
from scipy import sparse

from . import utils
from .. import _lib as L

__all__ = ["DenoisingAutoencoder"]


def pinned_copy(arr):
    """NumPy array -> pinned torch tensor (plain tensor without a GPU), filled by a single-threaded memcpy.  torch's own
    ``pin_memory()`` / ``copy_`` wake the whole intra-op OpenMP pool above 32 K elements; inside a container with a CPU quota the
    spinning workers (128 here) exhaust the quota and the process -- kernel launches included -- is throttled for whole 100 ms
    scheduler periods (measured: fit(rng='numpy') 0.7 M instead of 4.5 M samples/s).  The buffer comes from torch's caching host
    allocator, so it is recycled only after the asynchronous upload that reads it has completed."""
    import torch
    arr = np.ascontiguousarray(arr)
    t = torch.empty(arr.shape, dtype=torch.from_numpy(arr[:0]).dtype, pin_memory=torch.cuda.is_available())
    t.numpy()[...] = arr
    return t


_COPY_STREAMS = {}


def upload_ahead(draw, device, names):
    """Feeder thread: start the host -> device copies of the epoch's PINNED tensors `names` (keys of `draw`) on a copy stream of `device`, one epoch
    ahead of their use.  A copy issued on the stepping stream itself costs the step that opens an epoch ~34 us (tools/region_trace.py: the copy engine
    and the compute queue hand over through two signals); issued here it has a whole epoch to land, and `uploaded()` on the stepping thread only
    orders the stream behind an event that completed long ago.  No-op without a GPU device."""
    import torch
    if device is None or torch.device(device).type != 'cuda' or not torch.cuda.is_available():
        return draw
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device())
    st = _COPY_STREAMS.get(key)
    if st is None:
        st = _COPY_STREAMS[key] = torch.cuda.Stream(device=device)
    dev = {}
    with torch.cuda.device(device), torch.cuda.stream(st):
        for n in names:
            if n in draw and hasattr(draw[n], 'is_pinned') and draw[n].is_pinned():
                dev[n] = draw[n].to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(st)
    draw['_dev'] = dev
    draw['_dev_event'] = ev
    return draw


def uploaded(draw, name, device):
    """Stepping thread: the device copy of draw[name] -- the one `upload_ahead` started (the current stream is ordered behind its event, the block is
    marked as used by this stream), else an asynchronous copy on the current stream."""
    import torch
    dev = draw.get('_dev')
    if dev is not None and name in dev:
        cur = torch.cuda.current_stream(dev[name].device)
        if not draw.get('_dev_waited'):
            cur.wait_event(draw['_dev_event'])
            draw['_dev_waited'] = True
        dev[name].record_stream(cur)
        return dev[name]
    return draw[name].to(device, non_blocking=True)


class _EpochFeeder(object):
    """Produces the per-epoch host draws one epoch AHEAD of the device, on a thread, in epoch order -- so the legacy global
    NumPy stream is consumed exactly as the reference consumes it (corruption of epoch e, shuffle of epoch e, corruption of
    epoch e+1, ...) while the draw of epoch e+1 (a 1.6 M-element ``rand`` at 8000 x 10000) overlaps the GPU's epoch e.
    NumPy releases the GIL inside its generators.  Nothing else may draw from ``np.random`` while a fit is running."""

    def __init__(self, draw_fn, n_epochs):
        import queue
        import threading
        self._q = queue.Queue(maxsize=2)
        self._stop = False

        def run():
            try:
                # (no torch CPU op runs on this thread: pinned_copy fills its buffers through NumPy, so the intra-op OpenMP pool is
                # never woken from here and the process-wide torch thread count is left alone)
                for e in range(n_epochs):
                    if self._stop:
                        return
                    self._q.put(("ok", draw_fn(e)))
            except BaseException as exc:          # surfaces in the training thread
                self._q.put(("err", exc))
        self._t = threading.Thread(target=run, name="dae-epoch-feeder", daemon=True)
        self._t.start()

    def get(self):
        tag, val = self._q.get()
        if tag == "err":
            raise val
        return val

    def close(self):
        self._stop = True
        try:
            while self._t.is_alive():
                self._q.get_nowait()
        except Exception:
            pass
        self._t.join(timeout=5)


class DenoisingAutoencoder(object):
    """Denoising autoencoder with tied weights, h = f(W x~ + b) - f(b), optional online triplet mining.
    The interface is sklearn-like (reference autoencoder.py:14-18)."""

    _STRATEGIES = ['batch_all', 'batch_hard', 'none']

    def __init__(self, algo_name='dae', model_name='dae', compress_factor=10, main_dir='dae/', enc_act_func='tanh',
                 dec_act_func='none', loss_func='mean_squared', num_epochs=10, batch_size=10,
                 xavier_init=1, opt='gradient_descent', learning_rate=0.01, momentum=0.5, corr_type='none',
                 corr_frac=0., verbose=True, verbose_step=5, seed=-1, alpha=1, triplet_strategy='batch_all',
                 *, precision='auto', rng='numpy', init_weights=None, device=None, data_parallel=False,
                 dp_grad_dtype='fp32', dp_mining='local', results_root='results/', plan_options=None, dp_exchange='auto'):
        self.algo_name = algo_name
        self.model_name = model_name
        self.compress_factor = compress_factor
        self.main_dir = main_dir
        self.enc_act_func = enc_act_func
        self.dec_act_func = dec_act_func
        self.loss_func = loss_func
        self.num_epochs = num_epochs
        self.batch_size = batch_size
        self.xavier_init = xavier_init
        self.opt = opt
        self.learning_rate = learning_rate
        self.momentum = momentum
        self.corr_type = corr_type
        self.corr_frac = corr_frac
        self.verbose = verbose
        self.verbose_step = verbose_step
        self.seed = seed
        self.alpha = alpha
        self.triplet_strategy = triplet_strategy
        self.precision = precision
        self.rng = rng
        self.init_weights = init_weights
        self.device = device
        self.data_parallel = data_parallel
        self.dp_grad_dtype = dp_grad_dtype       # 'fp32' | 'bf16': element type of the gradient in the reduce-scatter (data parallel)
        self.dp_mining = dp_mining               # data parallel + triplet strategy: 'local' (each rank mines its shard) | 'global'
        assert self.dp_mining in ('local', 'global')
        self.dp_exchange = dp_exchange           # data parallel: 'auto' (split-bf16 mode: one fp32 all-reduce + full optimizer step on every rank;
        assert self.dp_exchange in ('auto', 'sharded', 'allreduce')     # else the sharded exchange) | 'sharded' | 'allreduce'  (dp.make_exchange)
        self.results_root = results_root
        self.plan_options = dict(plan_options or {})   # code-path choices of the step plan (dae_plan_set_option): A/B runs, tests

        assert type(self.verbose_step) == int                      # reference :68
        assert self.verbose >= 0
        assert self.triplet_strategy in self._STRATEGIES           # reference :70
        assert self.precision == 'auto' or self.precision in L.PRECISIONS, self.precision
        self.precision_used = None if self.precision == 'auto' else self.precision    # what 'auto' resolved to (set by fit / load_model)
        assert self.rng in ('numpy', 'philox')

        if self.seed >= 0:
            np.random.seed(self.seed)                              # reference :72-73 (the TF seed has no analogue)

        self.models_dir, self.data_dir, self.tf_summary_dir, self.tsv_dir, self.plot_dir = self._create_data_directories()
        self.model_path = self.models_dir + self.model_name
        self.parameter_file = self.tf_summary_dir + 'parameter.txt'

        self.sparse_input = None
        self.n_components = None
        self.engine = None
        self.history = []            # per verbose epoch: dict of the means the reference prints
        self.train_time = 0.0
        self.samples_per_sec = None

    # ------------------------------------------------------------------ bookkeeping (reference :101-124, :544-564)
    def _create_data_directories(self):
        algo = self.algo_name if self.algo_name[-1] == '/' else self.algo_name + '/'
        main = self.main_dir if self.main_dir[-1] == '/' else self.main_dir + '/'
        self.main_dir = algo + main
        root = self.results_root + self.main_dir
        dirs = [root + 'models/', root + 'data/', root + 'logs/', root + 'data/tsv/', root + 'data/plot/']
        for d in dirs:
            os.makedirs(d, exist_ok=True)
        return tuple(dirs)

    _PARAM_KEYS = ['algo_name', 'model_name', 'compress_factor', 'main_dir', 'enc_act_func', 'dec_act_func', 'loss_func',
                   'num_epochs', 'batch_size', 'xavier_init', 'opt', 'learning_rate', 'momentum', 'corr_type',
                   'corr_frac', 'verbose', 'verbose_step', 'seed', 'alpha', 'triplet_strategy']

    def _write_parameter_to_file(self, restore):
        with open(self.parameter_file, 'a+' if restore else 'w') as fh:
            print('---------------------------------------', file=fh)
            for k in self._PARAM_KEYS:
                print('{}={}'.format(k, getattr(self, k)), file=fh)
            print('precision={}'.format(self.precision_used or self.precision), file=fh)
            print('rng={}'.format(self.rng), file=fh)

    # ------------------------------------------------------------------ model construction
    def _strategy_key(self):
        return self.triplet_strategy

    def _resolve_batch(self, n_rows):
        bs = self.batch_size
        if bs < 1.:
            bs = max(round(n_rows * bs), 1)                         # reference utils.py:47
        return int(bs)

    @staticmethod
    def _abs_max(data):
        """Largest |value| of a train / validation / transform input: ndarray, scipy sparse matrix, torch tensor, or a list / dict of those (the explicit
        triplet estimator's org / pos / neg)."""
        if data is None:
            return 0.0
        if isinstance(data, dict):
            data = list(data.values())
        if isinstance(data, (list, tuple)):
            return max([DenoisingAutoencoder._abs_max(d) for d in data] + [0.0])
        vals = data if isinstance(data, np.ndarray) else getattr(data, "data", None)
        if vals is None and hasattr(data, "abs") and hasattr(data, "max"):      # torch tensor
            return float(data.abs().max()) if data.numel() else 0.0
        if isinstance(vals, memoryview) or vals is None:
            vals = np.asarray(data)
        return float(np.max(np.abs(vals))) if np.size(vals) else 0.0

    def _resolve_precision(self, data=None):
        """precision='auto' (the default) resolves PER TRIPLET STRATEGY to the cheapest mode measured to hold the reference's loss curve (L.AUTO_BY_STRATEGY:
        1e-4 on every one of 100 steps of the frozen float32-oracle curves for 'none' / 'batch_all' -- 'f16x2d' / 'f16x2h', fp16 operand images with W and
        the operands each strategy is sensitive to kept as hi + lo; the oracle-derived envelope for 'batch_hard' -- 'f16x2h' again), with or without a train set
        to look at (fit, load_model -> transform: one arithmetic).  'f16x2' (round 5's default: faster, holds 20 steps, leaves 1e-4 at step 37 of c2 / 76 of
        c1), plain 'bf16' / 'f16' (faster still, outside the gate) must be asked for by name."""
        if self.precision != 'auto':
            return self.precision
        mode = L.auto_precision(self._strategy_key())
        # fp16 images hold |x| <= 65504 (and the corrupted values scale * x with them): count data far outside tf-idf / binary BoW takes the split-bf16
        # mode, whose images have the fp32 range -- the only input-dependent part of 'auto'
        if data is not None and L.PRECISIONS[mode][0] == "f16" and self._abs_max(data) > 1.0e4:
            return 'bf16x3'
        return mode

    def _check_storage_range(self, data):
        """A model whose engine stores fp16 images (built by load_model() with no data to look at, or asked for by name) refuses values an fp16 image
        cannot hold instead of encoding infinities: the caller re-creates it with precision='bf16x3' (fp32 range).  Every matrix handed to fit() /
        validation / transform passes through here (lists and dicts of matrices included)."""
        used = self.precision_used or self._resolve_precision(None)
        if L.PRECISIONS[used][0] != "f16":
            return
        top = self._abs_max(data)
        if top > 6.0e4:
            raise ValueError("values up to %.3g do not fit the fp16 operand images of precision=%r: use precision='bf16x3' (or 'fp32')" % (top, used))

    def _build_engine(self, n_features, max_batch, dp_world=1, data=None):
        from ..engine import Engine                                # raises loudly without a GPU / the library
        self.precision_used = self._resolve_precision(data)
        if data is not None:
            self._check_storage_range(data)                        # an fp16 mode asked for by name on data it cannot hold: refused, not saturated
        if self.opt not in L.OPT:
            raise ValueError("unknown optimizer %r (reference :444-475 silently builds no train step)" % (self.opt,))
        act = lambda a: a if a in ('sigmoid', 'tanh') else 'none'
        self.engine = Engine(n_features, self.n_components, max_batch,
                             dtype=self.precision_used, enc_act=act(self.enc_act_func), dec_act=act(self.dec_act_func),
                             loss_func=self.loss_func, opt=self.opt, learning_rate=self.learning_rate,
                             momentum=self.momentum, alpha=float(self.alpha), triplet=self._strategy_key(),
                             device=self.device, dp_world=dp_world,
                             grad_lo=(dp_world > 1 and self.dp_grad_dtype == 'bf16' and self.dp_exchange != 'allreduce'))   # (the all-reduce moves the flat fp32 gradient)
        for name, value in self.plan_options.items():
            self.engine.set_option(name, value)
        return self.engine

    def _initial_parameters(self, n_features):
        if self.init_weights is not None:
            iw = self.init_weights
            W0 = iw[0] if isinstance(iw, (tuple, list)) else iw
            bh0 = iw[1] if isinstance(iw, (tuple, list)) and len(iw) > 1 else None
            bv0 = iw[2] if isinstance(iw, (tuple, list)) and len(iw) > 2 else None
            assert np.shape(W0) == (n_features, self.n_components), (np.shape(W0), (n_features, self.n_components))
            return W0, bh0, bv0
        # reference :365-367; a private stream seeded like tf.set_random_seed(seed) would be (:74), never the global one
        rng = np.random.RandomState(self.seed) if self.seed >= 0 else None
        return utils.xavier_init(n_features, self.n_components, self.xavier_init, rng=rng), None, None

    # ------------------------------------------------------------------ fit
    def fit(self, train_set, validation_set=None, train_set_label=None, validation_set_label=None,
            restore_previous_model=False):
        """Fit the model to the data (reference :126-156).  ``train_set``: ndarray (dense path) or scipy
        sparse matrix; labels are required iff ``triplet_strategy != 'none'``."""
        if self.triplet_strategy != 'none':
            assert train_set_label is not None
        if train_set_label is not None:
            assert train_set.shape[0] == len(train_set_label)
        if validation_set is not None and validation_set_label is not None:
            assert validation_set.shape[0] == len(validation_set_label)

        n_features = train_set.shape[1]
        if hasattr(self, '_sp_range'):
            del self._sp_range             # salt-and-pepper min / max belong to the previous fit's data
        self.sparse_input = not isinstance(train_set, np.ndarray)
        self.n_components = int(np.floor(n_features / self.compress_factor))
        batch = self._resolve_batch(train_set.shape[0])
        self._val_engine = None            # a cached validation engine would keep the previous fit's data and configuration

        world, rank = self._dist()
        local_batch = -(-batch // world)
        eng = self._build_engine(n_features, local_batch, dp_world=world, data=train_set)
        self._exchange = None
        if self.sparse_input:
            eng.upload_csr(train_set)
        else:
            eng.upload_dense(train_set)
        W0, bh0, bv0 = self._initial_parameters(n_features)
        if world > 1:
            from .. import dp
            # every rank must draw the SAME corruption masks and shuffles (each rank slices its shard out of one global
            # permutation): with seed >= 0 the constructor seeded the legacy stream identically everywhere; with seed < 0
            # rank 0's entropy is shared.  Parameters (W0 and any injected biases) are rank 0's.
            if self.seed < 0:
                shared = dp.broadcast_array(np.array([np.random.randint(0, 2 ** 31 - 1)], np.int64))
                np.random.seed(int(shared[0]))
            W0 = dp.broadcast_array(np.asarray(W0, np.float32))
            bh0 = dp.broadcast_array(np.zeros(self.n_components, np.float32) if bh0 is None else np.asarray(bh0, np.float32))
            bv0 = dp.broadcast_array(np.zeros(n_features, np.float32) if bv0 is None else np.asarray(bv0, np.float32))
            if self.triplet_strategy != 'none' and self.dp_mining == 'local' and rank == 0:
                import warnings
                warnings.warn("data_parallel with triplet_strategy=%r mines WITHIN each rank's shard of the mini-batch "
                              "(SURVEY 8e mode ii): this is the reference objective at batch size B/world on a different row "
                              "order, not the reference at batch size B; dp_mining='global' mines over the all-gathered batch" %
                              (self.triplet_strategy,))
        eng.set_params(W0, bh0, bv0)
        if restore_previous_model:
            self._restore(self.model_path)
        if rank == 0:
            self._write_parameter_to_file(restore_previous_model)
        if world > 1:
            from .. import dp
            self._exchange = dp.make_exchange(eng, grad_dtype=self.dp_grad_dtype, kind=self.dp_exchange)
            self._miner = None
            if self.triplet_strategy != 'none' and self.dp_mining == 'global':
                self._miner = dp.GlobalMiner(eng, self.triplet_strategy, float(self.alpha), batch)

        self._train_model(train_set, validation_set, train_set_label, validation_set_label)
        if world > 1:
            self._exchange.gather_master()  # the fp32 masters are sharded over the ranks during training
            self._exchange.gather_slots()   # ... and so are the optimizer slots of W (the checkpoint below saves them)
        if rank == 0:                      # one writer: every rank holds identical parameters
            self._save(self.model_path)
        if world > 1:
            from .. import dp
            dp.barrier()                   # nobody reads the checkpoint / artefacts before rank 0 has written them
        return None

    def _dist(self):
        if not self.data_parallel:
            return 1, 0
        from .. import dp
        return dp.world_size(), dp.rank()

    def _train_model(self, train_set, validation_set, train_set_label, validation_set_label):
        """Epoch loop (reference :175-204).  The host randomness of epoch e+1 (keep decisions of the whole set, then the shuffle
        -- the reference's draw order, :218-220) is produced by a feeder thread while the GPU runs epoch e."""
        import torch
        eng = self.engine
        N = train_set.shape[0]
        batch = self._resolve_batch(N)
        world, rank = self._dist()
        label_ids = None
        if train_set_label is not None:
            from .triplet_loss_utils import _labels_to_ids
            label_ids = _labels_to_ids(train_set_label)
        n_batches = -(-N // batch)
        self._stats = torch.zeros((max(self.num_epochs, 1), n_batches, L.STATS_STRIDE), dtype=torch.float32,
                                  device=eng.device)
        self._epoch_seconds = []
        # single GPU + a mining strategy: every mini-batch is handed over class-sorted (utils.class_sort_batches); under data
        # parallel the ranks take contiguous shards of a batch, which must stay a random sample of it
        sort_batch = batch if (world == 1 and label_ids is not None and self.triplet_strategy != 'none') else None
        dev = self.engine.device
        feeder = _EpochFeeder(lambda e: upload_ahead(self._stage_epoch(self._draw_epoch(train_set, e), label_ids, sort_batch), dev,
                                                     ('order_labels_t', 'order_t', 'labels_t', 'bits_t')), self.num_epochs)
        t_fit = time.time()
        t_first = None
        i = -1
        try:
            for i in range(self.num_epochs):
                t0 = time.time()
                self._run_train_step(train_set, label_ids, i, batch, world, rank, feeder.get())
                if (i + 1) % self.verbose_step == 0:
                    torch.cuda.synchronize()
                    self.train_time = time.time() - t0
                    self._run_validation_error_and_summaries(i + 1, validation_set, validation_set_label)
                self._epoch_seconds.append(time.time() - t0)
                if i == 0:
                    torch.cuda.synchronize()
                    t_first = time.time()
            else:
                if self.num_epochs != 0 and (i + 1) % self.verbose_step != 0:
                    torch.cuda.synchronize()
                    self.train_time = self._epoch_seconds[-1]
                    self._run_validation_error_and_summaries(i + 1, validation_set, validation_set_label)
        finally:
            feeder.close()
        torch.cuda.synchronize()
        t_end = time.time()
        if self.num_epochs > 1 and t_end > t_first:        # N * timed epochs / wall, first epoch excluded as warm-up
            self.samples_per_sec = N * (self.num_epochs - 1) / (t_end - t_first)
        elif self.num_epochs > 0 and t_end > t_fit:
            self.samples_per_sec = N * self.num_epochs / (t_end - t_fit)

    def _draw_epoch(self, train_set, epoch, n_shuffle=None):
        """Host randomness of one epoch in the reference's order: corrupt the WHOLE set, then shuffle (:218-220).  Runs on the
        feeder thread; everything it returns is host data (the uploads happen on the training thread)."""
        draw = dict(kind=self.corr_type)
        if self.corr_type == 'masking':
            if self.rng == 'numpy':
                # the legacy stream, continued natively (bit-identical to np.random.rand(nnz) >= v, utils.py:111, resp. the
                # np.random.choice([0, 1], size, p=[v, 1 - v]) of the dense path, utils.py:108)
                if self.sparse_input:
                    bits = utils.masking_keep_bits(self.engine.csr["nnz"], self.corr_frac)
                else:
                    bits = utils.masking_keep_bits(train_set.shape[0] * train_set.shape[1], utils.dense_masking_threshold(self.corr_frac))
                draw['bits'] = bits.view(np.int32)
        elif self.corr_type == 'salt_and_pepper':
            v = int(np.round(self.corr_frac * train_set.shape[1]))                           # reference :187
            if self.rng == 'philox' and self.sparse_input:
                draw['sp_v'] = v                                                             # flipped per batch on the device
            else:
                draw['xc'] = sparse.csr_matrix(utils.salt_and_pepper_noise(train_set, v))    # reference-exact host stream
        elif self.corr_type not in ('decay', 'none'):
            raise ValueError("unknown corr_type %r (reference :268 returns None and fails later)" % (self.corr_type,))
        n_rows = train_set.shape[0] if n_shuffle is None else n_shuffle
        draw['order'] = utils.epoch_permutation(n_rows)                                      # np.random.shuffle, after the corruption draws
        return draw

    @staticmethod
    def _stage_epoch(draw, label_ids, sort_batch=None):
        """Feeder thread: the epoch's host arrays as PINNED tensors (row order, labels in that order, keep bits), so the training
        thread's uploads are asynchronous copies instead of staged pageable ones."""
        if sort_batch and label_ids is not None:
            draw['order'] = utils.class_sort_batches(draw['order'], label_ids, sort_batch)
        order = draw['order']
        if label_ids is not None and np.asarray(label_ids).dtype == np.int32:
            # ONE pinned array [2 x N] = [row order | labels in that order]: the step that opens an epoch waits for ONE host -> device copy on its
            # stream instead of two (measured: +34 us on that step for the two copies, tools/region_trace.py)
            draw['order_labels_t'] = pinned_copy(np.stack([order.astype(np.int32), np.asarray(label_ids)[order]]))
        else:
            draw['order_t'] = pinned_copy(order.astype(np.int32))
            if label_ids is not None:
                draw['labels_t'] = pinned_copy(label_ids[order])
        if 'bits' in draw:
            draw['bits_t'] = pinned_copy(draw['bits'])
        return draw

    def _corruption_plan(self, draw, epoch):
        """Keyword arguments of Engine.train_step that realise this epoch's corruption (uploads what the feeder drew)."""
        import torch
        eng = self.engine
        if draw['kind'] == 'masking':
            if self.rng == 'philox':
                seed = self.seed if self.seed >= 0 else 0x5EED
                return dict(corr_mode=L.CORR_PHILOX_MASK, seed=seed, rng_stream=epoch, corr_frac=float(self.corr_frac))
            bits = uploaded(draw, 'bits_t', eng.device) if 'bits_t' in draw else torch.from_numpy(draw['bits']).to(eng.device, non_blocking=True)
            self._keep_bits = bits                                                           # keep alive while steps run
            return dict(corr_mode=L.CORR_KEEPBITS, keep_bits=bits)
        if draw['kind'] == 'decay':
            return dict(scale=1.0 - float(self.corr_frac))
        if draw['kind'] == 'salt_and_pepper':
            if 'sp_v' in draw:                                                               # device salt-and-pepper (rng='philox')
                return dict(_sp=(draw['sp_v'], self.seed if self.seed >= 0 else 0x5EED, epoch))
            from ..engine import Engine
            self._corrupted = Engine.to_device_csr(draw['xc'], eng.device)
            return dict(corrupted_csr=self._corrupted)
        return dict()

    def _run_train_step(self, train_set, label_ids, epoch, batch, world, rank, draw):
        """One epoch: corrupt, shuffle, then one fused device step per mini-batch (reference :206-246)."""
        import torch
        eng = self.engine
        N = train_set.shape[0]
        plan = self._corruption_plan(draw, epoch)
        order = draw['order']
        labels_dev = None
        if 'order_labels_t' in draw:
            both = uploaded(draw, 'order_labels_t', eng.device)
            order_dev, labels_dev = both[0], both[1]
        else:
            order_dev = uploaded(draw, 'order_t', eng.device) if 'order_t' in draw else torch.from_numpy(order.astype(np.int32)).to(eng.device, non_blocking=True)
            if label_ids is not None:
                labels_dev = uploaded(draw, 'labels_t', eng.device) if 'labels_t' in draw else torch.from_numpy(label_ids[order]).to(eng.device, non_blocking=True)
        stats = self._stats[epoch]
        shard_w = []
        sp = plan.pop('_sp', None)
        if sp is not None and not hasattr(self, '_sp_range'):
            # global minimum / maximum of the train set (utils.py:131-132); implicit zeros of a sparse matrix count
            d = train_set.data if train_set.nnz else np.zeros(1)
            full = train_set.nnz == train_set.shape[0] * train_set.shape[1]
            self._sp_range = (float(d.min() if full else min(d.min(), 0.0)), float(d.max() if full else max(d.max(), 0.0)))
        for b, start in enumerate(range(0, N, batch)):
            stop = min(N, start + batch)
            if world > 1:                                               # contiguous shard of every global batch
                from .. import dp
                lo, hi = dp.shard_bounds(start, stop, world, rank)
            else:
                lo, hi = start, stop
            rows = order_dev[lo:hi]
            labs = None if labels_dev is None else labels_dev[lo:hi]
            if sp is not None and hi > lo:
                plan['corrupted_csr'] = eng.salt_pepper_batch(rows, sp[0], self._sp_range[0], self._sp_range[1], sp[1], sp[2])
            if world > 1 and getattr(self, '_miner', None) is not None:
                # global-batch mining: encode my rows, mine over the all-gathered batch, resume; the ranks' gradients SUM to
                # the gradient of the reference cost at the global batch size
                if hi > lo:
                    eng.train_step(rows, labs, stats[b], phase=4, **plan)
                tl, fr, num = self._miner.mine(label_ids[order[start:stop]], start, stop)
                if hi > lo:
                    eng.train_step(rows, labs, stats[b], phase=5, **plan)
                else:
                    eng.zero_grads()
                    stats[b].zero_()
                stats[b, L.STAT_TRIPLET] = tl; stats[b, L.STAT_FRACTION] = fr; stats[b, L.STAT_NUM] = num
                self._exchange.step(grad_scale=1.0, grad_ready_after_dw=hi > lo)
            elif world > 1:
                # the global-batch mean is sum_r (rows_r / rows) * mean_r: every rank's gradient (and statistics) is weighted
                # by its share of the rows, so ragged tails (37 rows on 4 ranks = 10/10/10/7) and empty shards stay exact
                w = (hi - lo) / float(stop - start)
                shard_w.append(w)
                untouched = False                             # the W gradient is exactly what the step's dW kernel wrote
                if hi > lo:
                    eng.train_step(rows, labs, stats[b], phase=1, **plan)
                    if abs(w * world - 1.0) > 1e-12:
                        eng.grad.mul_(w * world)
                        if getattr(eng, "grad_lo", None) is not None:
                            eng.grad_lo.mul_(w * world)
                    else:
                        untouched = True
                else:
                    eng.zero_grads()
                    stats[b].zero_()
                # reduce-scatter (beside the step tail when the gradient is untouched) -> sharded optimizer -> all-gather -> unpack
                self._exchange.step(grad_scale=1.0 / world, grad_ready_after_dw=untouched)
            else:
                eng.train_step(rows, labs, stats[b], phase=3, **plan)
        if world > 1 and getattr(self, '_miner', None) is not None:      # AE legs add up (cw is globally normalised); triplet stats are global
            ae = stats[:, L.STAT_AE].clone()
            dp.allreduce_sum_(ae)
            stats[:, L.STAT_AE] = ae
            stats[:, L.STAT_COST] = ae + float(self.alpha) * stats[:, L.STAT_TRIPLET]
        elif world > 1:                                                 # statistics of the GLOBAL batches, on every rank
            wt = torch.tensor(shard_w, dtype=torch.float32, device=eng.device)[:, None]
            stats.mul_(wt)
            dp.allreduce_sum_(stats)

    # ------------------------------------------------------------------ reporting (reference :272-320)
    def epoch_stats(self, epoch):
        """Means over the epoch's batches of (cost, ae, triplet, fraction, num) -- what :283-294 prints."""
        s = self._stats[epoch - 1].cpu().numpy()
        return dict(cost=float(s[:, L.STAT_COST].mean()), ae=float(s[:, L.STAT_AE].mean()),
                    triplet=float(s[:, L.STAT_TRIPLET].mean()), fraction=float(s[:, L.STAT_FRACTION].mean()),
                    num=float(s[:, L.STAT_NUM].mean()), per_batch=s)

    def _run_validation_error_and_summaries(self, epoch, validation_set, validation_set_label):
        st = self.epoch_stats(epoch)
        rec = dict(epoch=epoch, seconds=self.train_time, **{k: st[k] for k in ('cost', 'ae', 'triplet', 'fraction', 'num')})
        main = self._dist()[1] == 0                                     # data parallel: rank 0 reports (statistics are global)
        if self.verbose == 1 and main:
            print('At step %d (%.2f seconds): ' % (epoch, self.train_time), end='')
            print('[Train Stat (average over past steps)] - ', end='')
            if self.triplet_strategy != 'none':
                print('Triplet: ', end='')
                print('Fraction=%.4f\t' % st['fraction'], end='')
                print('Number=%.2f\t' % st['num'], end='')
            print('Cost: ', end='')
            print('Overall=%.4f\t' % st['cost'], end='')
            if self.triplet_strategy != 'none':
                print('Autoencoder=%.4f\t' % st['ae'], end='')
                print('Triplet=%.4f\t' % st['triplet'], end='')
        if validation_set is None:
            if self.verbose == 1 and main:
                print()
            self.history.append(rec)
            return
        if getattr(self, '_exchange', None) is not None:
            self._exchange.gather_master()
        v = self._validation_forward(validation_set, validation_set_label)
        rec.update(val_cost=v[L.STAT_COST], val_ae=v[L.STAT_AE], val_triplet=v[L.STAT_TRIPLET])
        if self.verbose and main:
            print("[Validation Stat (at this step)] - Cost: ")
            print('Overall=%.4f' % v[L.STAT_COST], end='')
            if self.triplet_strategy != 'none':
                print('Autoencoder=%.4f\t' % v[L.STAT_AE], end='')
                print('Triplet=%.4f\t' % v[L.STAT_TRIPLET], end='')
            print()
        self.history.append(rec)

    def _forward_precision(self, data):
        """Precision of a forward-only engine over `data` (validation): the training precision; a validation set an fp16-storage engine cannot hold is
        refused like a train set (it used to skip the range check)."""
        used = self.precision_used or self._resolve_precision(data)
        if L.PRECISIONS[used][0] == "f16" and self._abs_max(data) > 6.0e4:
            raise ValueError("validation values up to %.3g do not fit the fp16 operand images of precision=%r: use precision='bf16x3' (or 'fp32')"
                             % (self._abs_max(data), used))
        return used

    def _validation_forward(self, validation_set, validation_set_label):
        """Forward pass of the whole validation set as ONE batch, uncorrupted (reference :300-312)."""
        import torch
        from ..engine import Engine
        from .triplet_loss_utils import _labels_to_ids
        nv, F = validation_set.shape
        if getattr(self, '_val_engine', None) is None or self._val_engine.Bmax < nv:
            act = lambda a: a if a in ('sigmoid', 'tanh') else 'none'
            self._val_engine = Engine(F, self.n_components, nv, dtype=self._forward_precision(validation_set), enc_act=act(self.enc_act_func),
                                      dec_act=act(self.dec_act_func), loss_func=self.loss_func, opt='gradient_descent',
                                      alpha=float(self.alpha), triplet=self._strategy_key(), device=self.device)
            if isinstance(validation_set, np.ndarray):
                self._val_engine.upload_dense(validation_set)
            else:
                self._val_engine.upload_csr(validation_set)
        ve = self._val_engine
        ve.set_params(*self.engine.get_params())
        labels = None
        if self.triplet_strategy != 'none':
            labels = torch.from_numpy(_labels_to_ids(validation_set_label)).to(ve.device)
        stats = torch.zeros(L.STATS_STRIDE, dtype=torch.float32, device=ve.device)
        ve.train_step(torch.arange(nv, dtype=torch.int32, device=ve.device), labels, stats, phase=2)
        return stats.cpu().numpy()

    # ------------------------------------------------------------------ inference / persistence
    def transform(self, data, name='train', save=False):
        """Encode ``data`` with the trained model: act(x W + bh) - act(bh)  (reference :479-505).
        Serves from the in-memory weights (the reference re-opens a session and restores the checkpoint)."""
        import torch
        from ..engine import Engine
        assert self.engine is not None, "fit() or load_model() first"
        eng = self.engine
        self._check_storage_range(data)
        n = data.shape[0]
        dev = eng.device
        out = torch.empty((n, eng.H), dtype=torch.float32, device=dev)
        if isinstance(data, np.ndarray):
            src = dict(dense=torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).to(dev))
        else:
            src = dict(csr=Engine.to_device_csr(data, dev))
        step = eng.Bmax
        for i0 in range(0, n, step):
            idx = torch.arange(i0, min(n, i0 + step), dtype=torch.int32, device=dev)
            eng.encode_rows(idx, out[i0:i0 + idx.numel()], **src)
        encoded = out.cpu().numpy()
        if save and self._dist()[1] == 0:
            np.save(self.data_dir + name, encoded)
            np.save(self.data_dir + 'weights', eng.get_params()[0])
        return encoded

    def _save(self, path):
        W, bh, bv = self.engine.get_params()
        state = {'enc-w': W, 'hidden-bias': bh, 'visible-bias': bv, 'adam_t': np.int64(self.engine.adam_t)}
        for k in ('s1', 's2'):
            t = getattr(self.engine, k)
            if t is not None:
                state['opt-' + k] = t.cpu().numpy()
        np.savez(path + '.npz', **state)

    def _restore(self, path):
        import torch
        z = np.load(path if path.endswith('.npz') else path + '.npz')
        self.engine.set_params(z['enc-w'], z['hidden-bias'], z['visible-bias'])
        for k in ('s1', 's2'):
            t = getattr(self.engine, k)
            if t is not None and ('opt-' + k) in z and z['opt-' + k].shape == tuple(t.shape):
                t.copy_(torch.from_numpy(z['opt-' + k]))
        self.engine.adam_t = int(z['adam_t']) if 'adam_t' in z else 0

    def load_model(self, shape, model_path):
        """Restore a trained model: shape = (n_features, n_components)  (reference :507-527)."""
        self.n_components = int(shape[1])
        self._build_engine(int(shape[0]), self._resolve_batch(1024) if self.batch_size >= 1 else 1024)
        self._restore(model_path)

    def get_model_parameters(self):
        """{'enc_w', 'enc_b', 'dec_b'} as NumPy arrays (reference :529-542)."""
        W, bh, bv = self.engine.get_params()
        return {'enc_w': W, 'enc_b': bh, 'dec_b': bv}
