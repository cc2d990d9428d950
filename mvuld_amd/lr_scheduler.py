"""``build_scheduler(config, optimizer, n_iter_per_epoch)`` -- the reference's per-iteration schedules
(mvuld/lr_scheduler.py:13-101) without timm: cosine = timm ``CosineLRScheduler(t_initial, t_mul=1, lr_min,
warmup_lr_init, warmup_t, cycle_limit=1, t_in_epochs=False)``; linear = the reference's own LinearLRScheduler; step =
timm ``StepLRScheduler(decay_t, decay_rate, warmup...)``.  ``step_update(num_updates)`` is called once per optimizer
update (main_bigvul.py:342); ``state_dict``/``load_state_dict`` for checkpoints."""
import math


class _Scheduler:
    def __init__(self, optimizer, warmup_t, warmup_lr_init):
        self.optimizer = optimizer
        self.warmup_t, self.warmup_lr_init = warmup_t, warmup_lr_init
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.base_values = [g["initial_lr"] for g in optimizer.param_groups]
        if warmup_t:
            self.warmup_steps = [(v - warmup_lr_init) / warmup_t for v in self.base_values]
            self._set([warmup_lr_init for _ in self.base_values])
        else:
            self.warmup_steps = [1 for _ in self.base_values]

    def _set(self, values):
        for g, v in zip(self.optimizer.param_groups, values):
            g["lr"] = v

    def _get_lr(self, t):
        raise NotImplementedError

    def get_update_values(self, num_updates):
        return self._get_lr(num_updates)

    def step_update(self, num_updates, metric=None):
        v = self.get_update_values(num_updates)
        if v is not None:
            self._set(v)

    def step(self, epoch, metric=None):          # t_in_epochs=False: per-epoch stepping is a no-op
        return None

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


class CosineLRScheduler(_Scheduler):
    def __init__(self, optimizer, t_initial, lr_min=0.0, warmup_t=0, warmup_lr_init=0.0):
        super().__init__(optimizer, warmup_t, warmup_lr_init)
        self.t_initial, self.lr_min = t_initial, lr_min

    def _get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        if t < self.t_initial:            # cycle_limit=1, t_mul=1: one cosine cycle over [0, t_initial)
            return [self.lr_min + 0.5 * (v - self.lr_min) * (1 + math.cos(math.pi * t / self.t_initial)) for v in self.base_values]
        return [self.lr_min for _ in self.base_values]


class LinearLRScheduler(_Scheduler):
    def __init__(self, optimizer, t_initial, lr_min_rate, warmup_t=0, warmup_lr_init=0.0):
        super().__init__(optimizer, warmup_t, warmup_lr_init)
        self.t_initial, self.lr_min_rate = t_initial, lr_min_rate

    def _get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        t = t - self.warmup_t
        total_t = self.t_initial - self.warmup_t
        return [v - ((v - v * self.lr_min_rate) * (t / total_t)) for v in self.base_values]


class StepLRScheduler(_Scheduler):
    def __init__(self, optimizer, decay_t, decay_rate=1.0, warmup_t=0, warmup_lr_init=0.0):
        super().__init__(optimizer, warmup_t, warmup_lr_init)
        self.decay_t, self.decay_rate = decay_t, decay_rate

    def _get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        return [v * (self.decay_rate ** (t // self.decay_t)) for v in self.base_values]


def build_scheduler(config, optimizer, n_iter_per_epoch):
    num_steps = int(config.TRAIN.EPOCHS * n_iter_per_epoch)
    warmup_steps = int(config.TRAIN.WARMUP_EPOCHS * n_iter_per_epoch)
    decay_steps = int(config.TRAIN.LR_SCHEDULER.DECAY_EPOCHS * n_iter_per_epoch)
    name = config.TRAIN.LR_SCHEDULER.NAME
    if name == 'cosine':
        return CosineLRScheduler(optimizer, t_initial=num_steps, lr_min=config.TRAIN.MIN_LR,
                                 warmup_lr_init=config.TRAIN.WARMUP_LR, warmup_t=warmup_steps)
    if name == 'linear':
        return LinearLRScheduler(optimizer, t_initial=num_steps, lr_min_rate=0.01, warmup_lr_init=config.TRAIN.WARMUP_LR,
                                 warmup_t=warmup_steps)
    if name == 'step':
        return StepLRScheduler(optimizer, decay_t=decay_steps, decay_rate=config.TRAIN.LR_SCHEDULER.DECAY_RATE,
                               warmup_lr_init=config.TRAIN.WARMUP_LR, warmup_t=warmup_steps)
    return None
