"""One training iteration recorded as a HIP graph and replayed.

The iteration of the reference's trainers (/root/reference/train.py:139-146: ``optimizer.zero_grad()``,
``output = model(image)``, ``loss = criterion(output, target)``, ``loss.backward()``, ``optimizer.step()``) is about
1 400 kernel launches for DeepLabv3+ ResNet-101; issued one by one from Python the host can fall behind the GPU.
``GraphedTrainStep`` runs the iteration a few times on a side stream (library algorithm selection, lazy state),
records it once into a HIP graph with the batch in fixed buffers, and every later call copies the new batch into
those buffers and replays the graph: one host call per iteration, the same kernels in the same order.

What stays on the host between replays is what the reference does on the host: the learning-rate schedule
(``LR_Scheduler(optimizer, i, epoch, best_pred)`` writes ``param_groups[k]['lr']``, train.py:143).  The recorded
optimizer launch of ``segk.optim.SGD`` reads its hyper-parameters from device memory, refreshed by
``optimizer.sync_hyper()`` before each replay.  Random draws (dropout masks) advance with every replay through
PyTorch's graph-aware generator state.  Batch-norm running statistics, momentum buffers and weights are updated in
place by the replayed kernels, so ``state_dict()`` / checkpoints see them as after eager iterations.

Data-parallel jobs: exchanges through the mailboxes of segk.parallel (batch-norm statistics, loss sums) are launches
of this library and are recorded like any other.  Gradient buckets that travel through torch.distributed (RCCL) stay
outside: the iteration is then recorded as TWO graphs - everything up to the packed buckets, and the optimizer
launch - with ``buckets.exchange_flat()`` issued between their replays (three host calls per iteration instead of
some 1 400; the all-reduce does not overlap the backward pass).

Restrictions: the shapes of the batch are fixed at construction; a step whose batch norms or loss exchange through
torch.distributed collectives is refused (run those steps eagerly, or connect the mailboxes); the model must not
synchronise with the host inside the iteration (``.item()``, printing a loss) - read the returned loss tensor after
the call instead.
"""
import torch
import torch.distributed as dist


class GraphedTrainStep:
    def __init__(self, model, criterion, optimizer, image, target, warmup=3, loss_fn=None, buckets=None):
        """image / target: one example batch on the GPU (its values are used for the warm-up iterations).
        loss_fn(image, target) -> loss replaces ``criterion(model(image), target)`` when given (e.g.
        ``model.forward_loss``).  `warmup` eager iterations run first; they update the weights like any iteration.
        buckets: the segk.parallel.GradientBuckets of a data-parallel job; the iteration then ends with
        ``optimizer.step(flat_grads=buckets.reduce_flat())``."""
        from . import parallel
        # two graphs with the bucket all-reduce between them when only the buckets are on torch.distributed
        self.split = not parallel.step_is_capturable()
        if self.split and not (parallel.mailbox_active() and buckets is not None):
            raise RuntimeError("GraphedTrainStep: a step whose batch norms and loss exchange through torch.distributed "
                               "collectives (world size %d) is not recorded; run it eagerly, or connect the mailboxes "
                               "(segk.parallel.connect_mailboxes) and pass the GradientBuckets" % dist.get_world_size())
        if not image.is_cuda:
            raise RuntimeError("GraphedTrainStep: the batch must live on the GPU")
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.loss_fn = loss_fn
        self.buckets = buckets
        self.image = image.detach().clone()
        self.target = target.detach().clone()
        self.stream = torch.cuda.Stream(device=image.device)
        self._sync_hyper = getattr(optimizer, "sync_hyper", None)
        self._after_replay = getattr(optimizer, "after_graph_replay", None)
        self._lr_at_record = [float(g["lr"]) for g in optimizer.param_groups]
        self.stream.wait_stream(torch.cuda.current_stream(image.device))
        with torch.cuda.stream(self.stream):
            for _ in range(max(int(warmup), 1)):   # at least one: autograd's accumulation nodes must be born here
                self._iteration()
        torch.cuda.current_stream(image.device).wait_stream(self.stream)
        torch.cuda.synchronize(image.device)
        if self._sync_hyper is not None:
            self._sync_hyper()
            torch.cuda.synchronize(image.device)
        optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        self.graph_step = None
        if not self.split:
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.loss = self._iteration()
        else:
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.loss = self._forward_backward()
                self.buckets.pack_flat()
            self.graph_step = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_step, stream=self.stream, pool=self.graph.pool()):
                self.optimizer.step(flat_grads=self.buckets)

    def _forward_backward(self):
        self.optimizer.zero_grad(set_to_none=True)
        if self.loss_fn is not None:
            loss = self.loss_fn(self.image, self.target)
        else:
            loss = self.criterion(self.model(self.image), self.target)
        loss.backward()
        return loss.detach()

    def _iteration(self):
        loss = self._forward_backward()
        if self.split:
            self.optimizer.step(flat_grads=self.buckets.pack_flat().exchange_flat())
        elif self.buckets is not None:
            self.optimizer.step(flat_grads=self.buckets.reduce_flat())
        else:
            self.optimizer.step()
        return loss

    def __call__(self, image, target):
        """One iteration on the given batch (same shapes and dtypes as at construction).  Returns the loss tensor of
        this iteration; it is overwritten by the next call."""
        if image.shape != self.image.shape or target.shape != self.target.shape:
            raise ValueError("GraphedTrainStep: batch of shape %s / %s, recorded for %s / %s"
                             % (tuple(image.shape), tuple(target.shape), tuple(self.image.shape), tuple(self.target.shape)))
        if image.data_ptr() != self.image.data_ptr():
            self.image.copy_(image, non_blocking=True)
        if target.data_ptr() != self.target.data_ptr():
            self.target.copy_(target, non_blocking=True)
        if self._sync_hyper is not None:
            self._sync_hyper()
        elif [float(g["lr"]) for g in self.optimizer.param_groups] != self._lr_at_record:
            raise RuntimeError("GraphedTrainStep: the learning rate changed, but this optimizer's recorded launch keeps "
                               "the value it was recorded with; use segk.optim.SGD")
        self.graph.replay()
        if self.graph_step is not None:
            self.buckets.exchange_flat()
            self.graph_step.replay()
        if self._after_replay is not None:
            self._after_replay()
        return self.loss
