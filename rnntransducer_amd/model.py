"""RNNTransducer — the LightningModule surface of the reference's model.py on the MI355X HIP path.

Same constructor `(prednet_params, transnet_params, jointnet_params, args)`, same `forward`, `training_step`
(7-tuple batch of dataloader.py:49) and `configure_optimizers` (AdamW + OneCycleLR per step, model.py:110-126),
same attribute `jointnet` and therefore the same state_dict keys.  Inherits
pytorch_lightning.LightningModule when that package is importable, torch.nn.Module otherwise.

Differences, all inside the hot path:
  * `training_step` runs the FUSED joint + RNN-T loss (`JointNet.loss`): the (B,T,U+1,V) logits tensor the
    reference builds at model.py:56 is never materialised.  `forward()` still returns it on request.
  * blank/pad id comes from `args.blank_token_id` / `prednet_params["pad_token_id"]` (default 0, as in the shipped
    config.json:37,40) instead of loading a tokenizer (model.py:24-26): tokenizer and WER/CER are validation-side
    and out of scope.
  * `validation_step` stays on the GPU (the reference moves the module to the CPU at model.py:65-72 because its
    decode is a host loop): fused loss + on-device greedy search; it returns token ids, and texts only if a
    tokenizer object was attached as `self.tokenizer`.
"""
from argparse import Namespace

import torch

from .loss import RNNTLoss
from .networks import JointNet

try:  # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    pl = None
    _Base = torch.nn.Module


class RNNTransducer(_Base):
    def __init__(self, prednet_params: dict, transnet_params: dict, jointnet_params: dict, args: Namespace):
        super().__init__()
        if pl is not None:
            self.save_hyperparameters(prednet_params, transnet_params, jointnet_params, args)
        self.args = args
        self._ctor_args = {"prednet_params": dict(prednet_params), "transnet_params": dict(transnet_params),
                           "jointnet_params": dict(jointnet_params), "args": args}
        prednet_params = dict(prednet_params)
        blank = getattr(args, "blank_token_id", None)
        if blank is None:
            blank = prednet_params.get("pad_token_id", 0)
        self.blank_token_id = int(blank)
        prednet_params["pad_token_id"] = self.blank_token_id  # model.py:26
        self.jointnet = JointNet(dict(transnet_params), prednet_params, **jointnet_params)
        # model.py:28-39 picks torchaudio (precision 16) or warp-transducer; both are this one HIP module here
        self.rnnt_loss = RNNTLoss(blank=self.blank_token_id, reduction="mean")

    def forward(self, input_audios, audio_lengths, input_texts, text_lengths):
        # the reference hands these two as python lists (dataloader.py:20,37): validating them costs no device sync
        if isinstance(audio_lengths, (list, tuple)) and (max(audio_lengths) > input_audios.size(1) or min(audio_lengths) < 1):
            raise ValueError(f"audio_lengths must lie in [1, {input_audios.size(1)}]")
        if isinstance(text_lengths, (list, tuple)) and (max(text_lengths) > input_texts.size(1) or min(text_lengths) < 1):
            raise ValueError(f"text_lengths must lie in [1, {input_texts.size(1)}]")
        return self.jointnet(input_audios, audio_lengths, input_texts, text_lengths)

    # ---- checkpoint wire format -------------------------------------------------------------------
    @staticmethod
    def read_reference_checkpoint(path: str) -> dict:
        """Reads a Lightning .ckpt as the reference trainer writes it (train.py:31-37 ModelCheckpoint; `save_hyperparameters`
        at model.py:22 stores the three parameter dicts AND the `argparse.Namespace` under `hyper_parameters`).  Weights-only
        unpickling: nothing from the file is executed; `argparse.Namespace` — a plain attribute container — is the one
        extra class allowed."""
        with torch.serialization.safe_globals([Namespace]):
            return torch.load(path, map_location="cpu", weights_only=True)

    def load_reference_checkpoint(self, path: str, strict: bool = True):
        """Loads the `state_dict` (keys `jointnet.*`) of a reference-written Lightning .ckpt into this module."""
        blob = self.read_reference_checkpoint(path)
        sd = blob.get("state_dict", blob)
        return self.load_state_dict({k: v for k, v in sd.items() if k.startswith("jointnet.")}, strict=strict)

    @classmethod
    def from_reference_checkpoint(cls, path: str, **overrides):
        """`RNNTransducer.load_from_checkpoint(path, prednet_params=..., ...)` of inference.py:19-25 without Lightning:
        constructor arguments come from the file's `hyper_parameters` unless overridden."""
        blob = cls.read_reference_checkpoint(path)
        hp = dict(blob.get("hyper_parameters", {}))
        hp.update(overrides)
        model = cls(hp["prednet_params"], hp["transnet_params"], hp["jointnet_params"], hp["args"])
        sd = blob["state_dict"]
        model.load_state_dict({k: v for k, v in sd.items() if k.startswith("jointnet.")})
        return model

    def save_reference_checkpoint(self, path: str, epoch: int = 0, global_step: int = 0, optimizer=None) -> None:
        """Writes a Lightning-1.8-shaped .ckpt the reference's `load_from_checkpoint` / `--resume_from_checkpoint` accept:
        `state_dict` (CPU tensors, keys `jointnet.*`), `hyper_parameters` = what model.py:22 saves (three dicts + the
        Namespace), epoch / global_step, `optimizer_states` in torch's format when an optimizer is given."""
        jp = dict(self._ctor_args["jointnet_params"])
        blob = {
            "epoch": int(epoch), "global_step": int(global_step), "pytorch-lightning_version": "1.8.0",
            "state_dict": {k: v.detach().cpu().clone() for k, v in self.state_dict().items()},
            "hyper_parameters": {"prednet_params": dict(self._ctor_args["prednet_params"]),
                                 "transnet_params": dict(self._ctor_args["transnet_params"]), "jointnet_params": jp,
                                 "args": self._ctor_args["args"]},
            "hparams_name": "kwargs",
        }
        if optimizer is not None:
            osd = optimizer.state_dict()
            osd["state"] = {k: {n: (t.detach().cpu().clone() if isinstance(t, torch.Tensor) else t) for n, t in st.items()}
                            for k, st in osd["state"].items()}
            blob["optimizer_states"] = [osd]
        torch.save(blob, path)

    def training_step(self, batch, batch_idx):
        assert not getattr(self.args, "move_metrics_to_cpu", False), "DDP only (model.py:53)"
        input_audios, audio_lengths, tensor_audio_lengths, input_texts, text_lengths, targets, target_lengths = batch
        loss = self.jointnet.loss(input_audios, tensor_audio_lengths, input_texts, targets, target_lengths,
                                  self.blank_token_id, reduction="mean",   # reduction="mean" (model.py:39), inside the library
                                  audio_lengths=audio_lengths if isinstance(audio_lengths, (list, tuple)) else None)
        if pl is not None and getattr(self, "_trainer", None) is not None:
            self.log("train_loss", loss, sync_dist=True)
        return {"loss": loss}

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        """model.py:62-79: loss + greedy search (max 3 symbols per frame).  `pred_tokens` is a list of B 1-D LongTensors
        (each what the reference's batch-1 recognize_greedy returns for that utterance), `label_tokens` the un-padded targets."""
        input_audios, audio_lengths, tensor_audio_lengths, input_texts, text_lengths, targets, target_lengths = batch
        nll = self.jointnet.loss(input_audios, tensor_audio_lengths, input_texts, targets, target_lengths, self.blank_token_id)
        was_training = self.jointnet.training
        self.jointnet.eval()
        try:
            pred = self.jointnet.recognize_greedy(input_audios, tensor_audio_lengths, self.blank_token_id, 3)
        finally:
            self.jointnet.train(was_training)
        pred = [p for p in pred] if isinstance(pred, list) else [pred[0]]
        u = target_lengths.tolist() if isinstance(target_lengths, torch.Tensor) else list(target_lengths)
        labels = [targets[b, :u[b]].long() for b in range(targets.size(0))]
        out = {"loss": nll.mean(), "pred_tokens": pred, "label_tokens": labels}
        tok = getattr(self, "tokenizer", None)
        if tok is not None:
            out["pred_texts"] = tok.batch_decode([p.tolist() for p in pred])
            out["label_texts"] = tok.batch_decode([l.tolist() for l in labels])
        return out

    def validation_epoch_end(self, validation_step_outputs):
        """model.py:81-108: mean validation loss + error rates over the epoch's validation_step outputs.  WER/CER when the
        steps carried texts (tokenizer attached), otherwise the token error rate on ids.  Returns the dict it logs."""
        from .metrics import char_error_rate, token_error_rate, word_error_rate
        out = {"val_loss": torch.stack([x["loss"].detach().reshape(()) for x in validation_step_outputs]).mean()}
        if validation_step_outputs and "pred_texts" in validation_step_outputs[0]:
            preds = [t for x in validation_step_outputs for t in x["pred_texts"]]
            labels = [t for x in validation_step_outputs for t in x["label_texts"]]
            out["val_wer"], out["val_cer"] = word_error_rate(preds, labels), char_error_rate(preds, labels)
        else:
            out["val_ter"] = token_error_rate([t for x in validation_step_outputs for t in x["pred_tokens"]],
                                              [t for x in validation_step_outputs for t in x["label_tokens"]])
        if pl is not None and getattr(self, "_trainer", None) is not None:
            for k, v in out.items():
                self.log(k, v.to(self.device) if isinstance(v, torch.Tensor) else v, sync_dist=True)
        return out

    def configure_optimizers(self):
        group = [{"params": [p for p in self.parameters()], "name": "OneCycleLR"}]
        trainer = getattr(self, "_trainer", None)
        if next(self.parameters()).is_cuda:
            # same optimiser, same hyper-parameters: parameters/gradients/moments flattened, one fused HIP step.
            # direct_grads (backward kernels add into the flat gradient views, no autograd accumulation kernels) is for loops
            # that use FlatAdamW.all_reduce_grads() as the DP exchange; under a Lightning trainer / DistributedDataParallel the
            # reducer needs autograd's accumulation hooks, so it stays off there.  args.direct_flat_grads overrides.
            from .optim import FlatAdamW
            direct = getattr(self.args, "direct_flat_grads", None)
            direct = (trainer is None) if direct is None else bool(direct)
            optimizer = FlatAdamW(group, lr=self.args.learning_rate, weight_decay=self.args.weight_decay, direct_grads=direct)
        else:
            optimizer = torch.optim.AdamW(group, lr=self.args.learning_rate, weight_decay=self.args.weight_decay)
        total = trainer.estimated_stepping_batches if trainer is not None else int(getattr(self.args, "total_steps"))
        scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=self.args.learning_rate, total_steps=total,
                                                        pct_start=self.args.warmup_ratio,
                                                        final_div_factor=self.args.final_div_factor)
        return {"optimizer": optimizer, "lr_scheduler": {"interval": "step", "scheduler": scheduler, "name": "AdamW"}}
