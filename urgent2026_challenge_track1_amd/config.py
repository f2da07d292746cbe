"""Training configuration: same attribute bag / CLI / yaml-overlay contract as the reference.

Mirrors ``baseline_code/config.py:6-72``: defaults (``:8-38``), yaml values override CLI flags and may add
new keys, ``train_tag`` := basename of the yaml (``:41-52``), one ``--flag`` per attribute with bools parsed
by ``str2bool`` (``:54-72``).  Extra keys: ``compute_dtype`` ("bf16" | "f16" | "f32") selects the MFMA operand type ("f16": IEEE-half operands in the forward contractions,
bf16 in the backward - the enhanced waveform then meets 1e-3 against the f32 reference arithmetic, bsrnn.BSRNNCore);
``unsupported_augmentation`` ("warn" | "raise" | "count") - see below.
"""
import argparse
import os

import yaml


class Config:
    def __init__(self, **kwargs):
        self.learning_rate = 1e-3
        self.batch_size = 2
        self.weight_decay = 1e-6
        self.adam_epsilon = 1e-8
        self.num_worker = 4
        self.num_train_epochs = 150
        self.device = "cuda"
        self.num_gpu = 1
        self.train_version = 0
        self.train_tag = "run_0"
        self.train_name = "baseline"
        self.val_check_interval = 50000
        self.save_top_k = 3
        self.resume = True
        self.seed = 1996
        self.gradient_clip = 0.5
        self.lr_step_size = 1
        self.lr_gamma = 0.85
        self.train_set_path = "none"
        self.train_set_dynamic_mixing = True
        self.valid_set_path = "none"
        self.init_from = "none"
        self.max_duration = 96000
        self.use_high_pass = True
        self.se_model = "bsrnn"
        self.config_file = "none"
        self.model_configs = None
        self.compute_dtype = "bf16"
        # what the trainer does when a dynamic-mixing recipe draws an augmentation the device simulator cannot apply (codec and the
        # wind-noise side-chain compressor need ffmpeg): "warn" = apply the rest, warn at the first one, report counters
        # with every log line; "raise" = stop at the first one; "count" = counters only
        self.unsupported_augmentation = "warn"
        for k, v in kwargs.items():
            setattr(self, k, v)

    def read_yaml(self):
        if self.config_file != "none":
            with open(self.config_file, "r", encoding="utf-8") as f:
                d = yaml.safe_load(f.read())
            for k, v in d.items():
                setattr(self, k, v)
            self.train_tag = os.path.basename(self.config_file).replace(".yaml", "")


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def config_parser(argv=None):
    cfg = Config()
    parser = argparse.ArgumentParser()
    for par, default in vars(cfg).items():
        if default is None:
            parser.add_argument("--%s" % par, default=None)
        else:
            parser.add_argument("--%s" % par, type=str2bool if isinstance(default, bool) else type(default),
                                default=default)
    return parser.parse_args(argv)
