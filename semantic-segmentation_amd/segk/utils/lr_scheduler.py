"""LR_Scheduler with the call shape of /root/reference/utils/lr_scheduler.py:32-73: poly / cos / step,
optional warm-up, first parameter group at lr and every further group at 10x lr.

The schedule formulas and the 1x / 10x group rule are those of the reference file, which carries the notice
"Created by: Hang Zhang, ECE Department, Rutgers University, zhang.hang@rutgers.edu, Copyright (c) 2017; this
source code is licensed under the MIT-style license found in the LICENSE file in the root directory of this
source tree" (PyTorch-Encoding); this file is an independent restatement of that interface."""
import math


class LR_Scheduler(object):
    def __init__(self, mode, base_lr, num_epochs, iters_per_epoch=0, lr_step=0, warmup_epochs=0, quiet=False):
        if mode not in ("poly", "cos", "step"):
            raise NotImplementedError(mode)
        if mode == "step":
            assert lr_step
        self.mode = mode
        self.lr = base_lr
        self.lr_step = lr_step
        self.iters_per_epoch = iters_per_epoch
        self.N = num_epochs * iters_per_epoch
        self.epoch = -1
        self.warmup_iters = warmup_epochs * iters_per_epoch
        self.quiet = quiet
        if not quiet:
            print("Using {} LR Scheduler!".format(self.mode))

    def __call__(self, optimizer, i, epoch, best_pred):
        T = epoch * self.iters_per_epoch + i
        if self.mode == "cos":
            lr = 0.5 * self.lr * (1 + math.cos(1.0 * T / self.N * math.pi))
        elif self.mode == "poly":
            lr = self.lr * pow((1 - 1.0 * T / self.N), 0.9)
        else:
            lr = self.lr * (0.1 ** (epoch // self.lr_step))
        if self.warmup_iters > 0 and T < self.warmup_iters:
            lr = lr * 1.0 * T / self.warmup_iters
        if epoch > self.epoch:
            if not self.quiet:
                print("\n=>Epoches %i, learning rate = %.4f, previous best = %.4f" % (epoch, lr, best_pred))
            self.epoch = epoch
        assert lr >= 0
        self._adjust_learning_rate(optimizer, lr)

    def _adjust_learning_rate(self, optimizer, lr):
        groups = optimizer.param_groups
        groups[0]["lr"] = lr
        for g in groups[1:]:
            g["lr"] = lr * 10
