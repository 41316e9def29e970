"""Optimiser descriptions with tf.keras' constructor arguments, as the reference builds them:

    SGD(learning_rate=ExponentialDecay(0.002, decay_steps=3*TR_STEPS, decay_rate=0.1), clipnorm=1, momentum=0.9)
                                                       lib/proposed_architectures.py:156-158 (B3_MTL)
    Adam(lr=1e-4) / Adam(lr=1e-3)                      :499-500 (Doukhan), :750-751 (Jang)
    SGD(learning_rate=ExponentialDecay(1e-3, 700, 0.1)) :572-574 (Papakostas)
    Nadam(learning_rate=0.002)                         DAFx12_Speech_Music_Detection_B3_MTL_v2.py:524-526 (sub-model fine-tune)

They are plain descriptions: `model.compile(optimizer=...)` copies them into the model, the update itself runs in
libsmh (smh_trainer_apply_f32 / smh_cnn_trainer_apply_f32).  Keras defaults: beta_1 0.9, beta_2 0.999, epsilon 1e-7.
"""
from __future__ import annotations


class ExponentialDecay:
    def __init__(self, initial_learning_rate, decay_steps, decay_rate, staircase=False):
        self.initial_learning_rate, self.decay_steps = float(initial_learning_rate), float(decay_steps)
        self.decay_rate, self.staircase = float(decay_rate), bool(staircase)

    def __call__(self, step):
        p = step / self.decay_steps
        if self.staircase:
            p = float(int(p))
        return self.initial_learning_rate * self.decay_rate ** p


class _Optimizer:
    kind = None

    def __init__(self, learning_rate, clipnorm, kwargs):
        if "lr" in kwargs:  # the reference writes optimizers.Adam(lr=...)
            learning_rate = kwargs.pop("lr")
        if kwargs:
            raise TypeError("%s: unsupported arguments %s" % (type(self).__name__, sorted(kwargs)))
        self.learning_rate = learning_rate
        self.clipnorm = None if clipnorm is None else float(clipnorm)

    def lr_at(self, step):
        return float(self.learning_rate(step)) if callable(self.learning_rate) else float(self.learning_rate)


class SGD(_Optimizer):
    kind = "sgd"

    def __init__(self, learning_rate=0.01, momentum=0.0, nesterov=False, clipnorm=None, **kwargs):
        super().__init__(learning_rate, clipnorm, kwargs)
        if nesterov:
            raise ValueError("SGD(nesterov=True) is not used by the reference and not built")
        self.momentum = float(momentum)


class Adam(_Optimizer):
    kind = "adam"

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, clipnorm=None, **kwargs):
        super().__init__(learning_rate, clipnorm, kwargs)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)


class Nadam(Adam):
    kind = "nadam"
