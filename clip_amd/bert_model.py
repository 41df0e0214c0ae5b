"""Host-side mirror of the reference's model classes, backed by the HIP engine.

Same names, call order and return shapes as /root/reference/sentclf/bert_model.py:
    model = BertSequenceMultilabelClassificationContext.from_pretrained(local_dir)   # test_load.py:21
    model.set_sep_token_id(102); model.set_n_context_sentences(2); model.set_task('multilabel')
    model.load_state_dict(torch.load(path)); model.to('cuda'); model.eval()          # apply_bert_unlabeled.py:72-80,100
    loss, logits = model(input_ids=..., attention_mask=..., token_type_ids=..., labels=...)

Differences, all forced by the environment: from_pretrained takes a LOCAL directory (config.json +
pytorch_model.bin | model.safetensors) — hub names cannot resolve offline; only inference is
implemented (no autograd, dropout is the identity as in model.eval()); the arithmetic runs in
libclip_amd.so on an MI355X and there is no CPU path.
"""
import ctypes as C
import os

import torch

from . import _lib
from .config import BertConfig
from .weights import state_dict_shapes

_IGNORED_KEYS = ("bert.embeddings.position_ids", "bert.embeddings.token_type_ids")


def _dev_i64(t, device):
    """a [B, S]-shaped argument of the padded call as contiguous int64 on the model's device (no-op for the tensors the
    HF tokenizer / the collators produce), None kept"""
    return None if t is None else t.to(device=device, dtype=torch.int64).contiguous()


class _EngineModel:
    """Shared plumbing: config, weights, engine lifetime, packed forward."""
    head_kind = 0
    mlm_head = False

    def __init__(self, config: BertConfig, max_tokens=65536, max_seqs=16384):
        config.validate_for_device()
        self.config = config
        self.num_labels = config.num_labels
        self.max_tokens, self.max_seqs = int(max_tokens), int(max_seqs)
        self.task = "multilabel"
        self.training = False
        self.device = torch.device("cpu")
        self._sd = None
        self._engine = None
        self.precision = "auto"             # what the caller asked for
        self.resolved_precision = None      # what the engine runs in ('auto' resolved by calibrate_precision)
        self.precision_report = None

    def set_precision(self, precision):
        """'auto' (default): the fastest 16-bit setting that MEASURES inside the parity criteria of BASELINE.json
        on a sample — logits within 2e-2 of the fp32 check mode and >= 99.9 % thresholded-label agreement
        (clip_amd/precision.py; the choice is printed and kept in .precision_report).  Or fixed: 'bf16' (bf16
        operands; within 2e-2 on benign weights, logits of +-2), 'f16' (fp16 operands at the same speed, 1/8 of
        bf16's operand rounding, at the price of fp16's range), 'f16r' (fp16 operands + fp32 residual stream:
        inside 2e-2 on the trained-like stress goldens, ~7 % slower) or 'fp32' (check mode: the same kernels on
        fp32 operands, within 1e-4; ~16x slower).  Rebuilds the engine on next use."""
        if precision not in ("auto",) + tuple(_lib.PRECISIONS):
            raise ValueError(f"precision must be 'auto', 'bf16', 'f16', 'f16r' or 'fp32', not {precision!r}")
        if precision != self.precision:
            self._drop_engine()
            self.precision = precision
            self.resolved_precision = None if precision == "auto" else precision
            self.precision_report = None
        return self

    def calibrate_precision(self, rows=None, label_threshs=None, log=None):
        """Resolve precision='auto' now, on `rows` (token-id rows of the caller's own input; default: synthetic rows
        of this model's layout).  No-op for a fixed precision or an already resolved model.  Returns the setting."""
        if self.precision == "auto" and self.resolved_precision is None:
            from . import precision as P
            if self.device.type != "cuda":
                raise RuntimeError("the model must be moved to an MI355X first (model.to('cuda')); "
                                   "clip_amd has no CPU execution path")
            if self._sd is None:
                raise RuntimeError("no weights loaded (from_pretrained / load_state_dict)")
            self.precision_report, clone = P.choose_precision(self, rows, label_threshs, log)
            self._drop_engine()
            self.resolved_precision = self.precision_report["selected"]
            if clone is not None:
                # the candidate that passed was built with this model's workspace size: adopt its engine
                if (clone.max_tokens, clone.max_seqs, clone.device) == (self.max_tokens, self.max_seqs, self.device):
                    self._engine, clone._engine = clone._engine, None
                    if hasattr(clone, "_trim_off"):
                        self._trim_off = clone._trim_off
                clone._drop_engine()
        return self.resolved_precision

    # ---- construction (test_load.py:21, apply_bert_unlabeled.py:65-70,78-79)
    @classmethod
    def from_pretrained(cls, path, config=None, **kwargs):
        if config is None:
            config = BertConfig.from_pretrained(path)
        model = cls(config, **{k: v for k, v in kwargs.items() if k in ("max_tokens", "max_seqs")})
        st = os.path.join(path, "model.safetensors")
        pt = os.path.join(path, "pytorch_model.bin")
        if os.path.isfile(st):
            from safetensors.torch import load_file
            model.load_state_dict(load_file(st))
        elif os.path.isfile(pt):
            model.load_state_dict(torch.load(pt, map_location="cpu", weights_only=True))
        else:
            raise OSError(f"no model.safetensors / pytorch_model.bin under {path!r}")
        return model

    def save_pretrained(self, path):
        from safetensors.torch import save_file
        self.config.save_pretrained(path)
        save_file({k: v.contiguous() for k, v in self.state_dict().items()},
                  os.path.join(path, "model.safetensors"))

    def load_state_dict(self, sd, strict=True):
        shapes = state_dict_shapes(self.config, self.mlm_head)
        sd = {k: v for k, v in sd.items() if k not in _IGNORED_KEYS}
        missing = [k for k in shapes if k not in sd]
        unexpected = [k for k in sd if k not in shapes]
        if self.head_kind == 0:  # the context head never reads the pooler
            missing = [k for k in missing if not k.startswith("bert.pooler.")]
        if any(k in sd for k in ("mlm_predictions.bias", "mlm_predictions.decoder.bias")):
            # one output bias: tied `mlm_predictions.bias` in old checkpoints, `decoder.bias` in new ones
            missing = [k for k in missing if k not in ("mlm_predictions.bias", "mlm_predictions.decoder.bias")]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {unexpected}")
        for k, v in sd.items():
            if k in shapes and tuple(v.shape) != tuple(shapes[k]):
                raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(shapes[k])}")
        self._sd = {k: sd[k].detach().to("cpu", torch.float32).contiguous() for k in shapes if k in sd}
        self._drop_engine()
        if self.precision == "auto":        # new weights: measure again
            self.resolved_precision = self.precision_report = None
        return missing, unexpected

    def state_dict(self):
        if self._sd is None:
            raise RuntimeError("no weights loaded")
        return dict(self._sd)

    # ---- setters (bert_model.py:23-32, 207-208)
    def set_task(self, task):
        self.task = task

    # ---- nn.Module-style lifetime
    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device != self.device:
            self._drop_engine()
            self.device = device
        return self

    def cuda(self):
        return self.to("cuda")

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("inference engine: training is out of scope")
        return self

    def _drop_engine(self):
        if self._engine is not None:
            _lib.load().clip_destroy(self._engine)
            self._engine = None

    def __del__(self):
        try:
            self._drop_engine()
        except Exception:
            pass

    def _ensure_engine(self):
        if self._engine is not None:
            return self._engine
        if self.device.type != "cuda":
            raise RuntimeError("the model must be moved to an MI355X first (model.to('cuda')); "
                               "clip_amd has no CPU execution path")
        if self._sd is None:
            raise RuntimeError("no weights loaded (from_pretrained / load_state_dict)")
        lib = _lib.load()
        self.calibrate_precision()          # precision='auto' not yet resolved: synthetic sample
        c = self.config
        cc = _lib.ClipConfig(c.vocab_size, c.hidden_size, c.num_hidden_layers, c.num_attention_heads,
                             c.intermediate_size, c.max_position_embeddings, c.type_vocab_size,
                             c.num_labels, c.layer_norm_eps, self.max_tokens, self.max_seqs)
        h = C.c_void_p()
        _lib.check(lib.clip_create(C.byref(cc), self.device.index, C.byref(h)))
        try:
            _lib.check(lib.clip_set_precision(h, _lib.PRECISIONS[self.resolved_precision]))
            for k, v in self._sd.items():
                _lib.check(lib.clip_load_weight(h, k.encode(), C.c_void_p(v.data_ptr()), v.numel()))
            _lib.check(lib.clip_finalize_weights(h))
            if os.environ.get("CLIP_AMD_FUSED_LN", "1") == "0":
                # the fused-LayerNorm GEMMs exchange row sums between co-resident workgroups: on a GPU
                # shared with other processes run the unfused flow (same results to the last few ulps)
                _lib.check(lib.clip_set_fused_layernorm(h, 0))
        except Exception:
            lib.clip_destroy(h)
            raise
        self._engine = h
        return h

    # ---- the packed fast path (used by clip_amd.scoring) ------------------------------------
    def focus_index(self, ids, cu_seqlens, check=True, return_err=False):
        raise NotImplementedError

    def forward_packed(self, ids, type_ids, pos_ids, cu_seqlens, max_seqlen, focus=None, check=True):
        """ids/type_ids/pos_ids int32 [T] and cu_seqlens int32 [n+1] on this model's device.
        Returns fp32 logits [n, num_labels]."""
        h = self._ensure_engine()
        lib = _lib.load()
        n, T = cu_seqlens.numel() - 1, ids.numel()
        sep_err = None
        if focus is None:
            # the [SEP]-count word is read AFTER the forward has been queued (one wait at the end of the call instead
            # of one in its middle; a bad row's focus index falls back to the row's first token, in bounds)
            focus, sep_err = self.focus_index(ids, cu_seqlens, check=False, return_err=True)
        logits = torch.empty((n, self.num_labels), dtype=torch.float32, device=ids.device)
        with torch.cuda.device(self.device):
            _lib.check(lib.clip_forward_packed(h, _lib.ptr(ids), _lib.ptr(type_ids), _lib.ptr(pos_ids),
                                               _lib.ptr(cu_seqlens), _lib.ptr(focus), n, T, int(max_seqlen),
                                               self.head_kind, _lib.ptr(logits), _lib.stream_ptr()))
            if check:      # 4-byte reads behind the forward; the bulk loop (clip_amd.scoring) polls once per chunk
                _lib.poll_errors(h)       # first: it also CLEARS the engine's word (an empty row sets both)
                if sep_err is not None:
                    self._raise_on_sep_count(sep_err)
        return logits

    def _raise_on_sep_count(self, err):
        n_ctx = getattr(self, "n_context_sentences", None)
        if n_ctx is not None and n_ctx >= 0 and int(err.item()) != 0:
            # reference: prints and returns None, then classifier(None) raises (bert_model.py:51-55,91)
            print("error when pulling out end of sentence indices: a row does not hold exactly "
                  f"{2 * n_ctx + 1} [SEP] tokens")
            raise TypeError("focus-sentence representation is None: wrong [SEP] count in a row")

    def _focus(self, ids, cu_seqlens, n_ctx, sep_id, check, return_err=False):
        lib = _lib.load()
        n = cu_seqlens.numel() - 1
        focus = torch.empty(n, dtype=torch.int32, device=ids.device)
        err = torch.zeros(1, dtype=torch.int32, device=ids.device)
        with torch.cuda.device(self.device):
            _lib.check(lib.clip_focus_index(_lib.ptr(ids), _lib.ptr(cu_seqlens), n, int(sep_id), int(n_ctx),
                                            _lib.ptr(focus), _lib.ptr(err), _lib.stream_ptr()))
        if check and n_ctx >= 0 and int(err.item()) != 0:
            # reference: prints and returns None, then classifier(None) raises (bert_model.py:51-55,91)
            print("error when pulling out end of sentence indices: a row does not hold exactly "
                  f"{2 * n_ctx + 1} [SEP] tokens")
            raise TypeError("focus-sentence representation is None: wrong [SEP] count in a row")
        return (focus, err) if return_err else focus

    # ---- the reference's padded forward ------------------------------------------------------
    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None,
                head_mask=None, inputs_embeds=None, labels=None, output_attentions=None,
                output_hidden_states=None, doc_positions=None):
        if head_mask is not None or inputs_embeds is not None or output_attentions or output_hidden_states:
            raise NotImplementedError("head_mask / inputs_embeds / attentions / hidden states are not on the scoring path")
        if doc_positions is not None and len(doc_positions) > 0:
            raise NotImplementedError("doc_position feature is not on the scoring path")
        if input_ids.device != self.device:
            raise RuntimeError(f"inputs on {input_ids.device}, model on {self.device}")
        if input_ids.dim() != 2:
            raise ValueError(f"input_ids must be [batch, width], got {tuple(input_ids.shape)}")
        B, S = input_ids.shape
        ids = _dev_i64(input_ids, self.device)
        mask = _dev_i64(attention_mask, self.device)
        tt = _dev_i64(token_type_ids, self.device)
        pos, pos_stride = None, 0
        if position_ids is not None:           # [S], [1, S] (broadcast, modeling_bert.py:83) or [B, S]
            pos = _dev_i64(position_ids, self.device).reshape(-1, S)
            if pos.shape[0] not in (1, B):
                raise ValueError(f"position_ids of shape {tuple(position_ids.shape)} for a batch of {B} x {S}")
            pos_stride = S if pos.shape[0] == B and B > 1 else 0
        for name, t in (("attention_mask", mask), ("token_type_ids", tt)):
            if t is not None and tuple(t.shape) != (B, S):
                raise ValueError(f"{name} of shape {tuple(t.shape)} for input_ids of shape {(B, S)}")
        h = self._ensure_engine()
        logits = torch.empty((B, self.num_labels), dtype=torch.float32, device=self.device)
        n_ctx = getattr(self, "n_context_sentences", None)
        if self.head_kind == 0 and (getattr(self, "sep_token_id", None) is None or n_ctx is None):
            raise AttributeError("call set_sep_token_id() and set_n_context_sentences() first")
        with torch.cuda.device(self.device):
            # ONE call: packs the rows on the device (the key-padding mask becomes the packing), finds the focus [SEP],
            # runs the encoder and returns the engine's error word (an empty row, a wrong [SEP] count, non-finite
            # logits); its two waits (token count, end of the forward) spin on words the device writes to host memory
            flags = C.c_int32(0)
            _lib.check(_lib.load().clip_forward_padded(
                h, _lib.ptr(ids), _lib.ptr(mask), _lib.ptr(tt), _lib.ptr(pos), pos_stride, B, S,
                int(getattr(self, "sep_token_id", None) or 0), int(n_ctx if n_ctx is not None else -1), self.head_kind,
                _lib.ptr(logits), None, C.byref(flags), _lib.stream_ptr()))
            _lib.raise_for_flags(flags)
        if labels is None:
            return (logits,)
        if self.task != "multilabel":
            raise NotImplementedError("only task='multilabel' is on the scoring path")
        lab = labels.to(self.device, torch.float32).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().clip_bce_loss(_lib.ptr(logits), _lib.ptr(lab), logits.numel(),
                                                 _lib.ptr(loss), _lib.stream_ptr()))
        return loss[0], logits

    __call__ = forward


class BertSequenceMultilabelClassificationContext(_EngineModel):
    """ref: sentclf/bert_model.py:8-103 — BERT over [CLS] ctx [SEP] .. focus [SEP] .. ctx [SEP],
    hidden state at the focus sentence's closing [SEP] -> Linear(H, num_labels)."""
    head_kind = 0

    def __init__(self, config, **kw):
        super().__init__(config, **kw)
        self.sep_token_id = None
        self.n_context_sentences = None

    def set_sep_token_id(self, sep_token_id):
        self.sep_token_id = int(sep_token_id)

    def set_n_context_sentences(self, n_context_sentences):
        self.n_context_sentences = int(n_context_sentences)

    def focus_index(self, ids, cu_seqlens, check=True, return_err=False):
        if self.sep_token_id is None or self.n_context_sentences is None:
            raise AttributeError("call set_sep_token_id() and set_n_context_sentences() first")
        return self._focus(ids, cu_seqlens, self.n_context_sentences, self.sep_token_id, check, return_err)


class BertContextMLMSwitch(BertSequenceMultilabelClassificationContext):
    """ref: sentclf/bert_model.py:106-186 — the BERT+Context encoder with an LM prediction head on every
    token and the switch classifier (2 classes) on the focus [SEP] row, as task-targeted pre-training
    evaluates it.  INFERENCE forward only (no autograd here): (total_loss, logits) as the reference returns
    them, total_loss = CrossEntropy(mlm_scores, mlm_labels) + CrossEntropy(logits, labels).
    CrossEntropyLoss ignores label -100, so the 768 -> vocab decoder runs on the labelled positions only
    (clip_mlm_head); `mlm_scores_at` gives the scores of bert_model.py:165 at those positions."""
    mlm_head = True

    def _ensure_engine(self):
        h = super()._ensure_engine()
        if not getattr(self, "_trim_off", None) == h.value:
            _lib.check(_lib.load().clip_set_trim_last_layer(h, 0))     # the LM head reads every token's last hidden state
            self._trim_off = h.value
        return h

    def expand_vocab_by_num(self, num):
        """ref: bert_model.py:116-135 — `num` more word embeddings (unit-norm random rows), a fresh
        decoder of the new size with zero bias; config.vocab_size grows."""
        sd = self.state_dict()
        H = self.config.hidden_size
        init = torch.nn.Embedding(num, H).weight.data.clone()
        init = init / init.norm(dim=1).unsqueeze(dim=1)
        sd["bert.embeddings.word_embeddings.weight"] = torch.cat([sd["bert.embeddings.word_embeddings.weight"], init])
        self.config.vocab_size += num
        dec = torch.nn.Linear(H, self.config.vocab_size, bias=False)
        sd["mlm_predictions.decoder.weight"] = dec.weight.data.clone()
        sd["mlm_predictions.bias"] = torch.zeros(self.config.vocab_size)
        sd["mlm_predictions.decoder.bias"] = torch.zeros(self.config.vocab_size)
        self.load_state_dict(sd)

    def _mlm_rows(self, mlm_labels, attention_mask):
        """packed token indices and labels of the positions CrossEntropyLoss does not ignore"""
        B, S = mlm_labels.shape
        mask = torch.ones((B, S), dtype=torch.bool, device=mlm_labels.device) if attention_mask is None \
            else attention_mask.to(torch.bool)
        packed_index = (torch.cumsum(mask.reshape(-1).to(torch.int64), 0) - 1).reshape(B, S)
        sel = mlm_labels != -100
        if bool((sel & ~mask).any()):
            raise NotImplementedError("an MLM label on a masked-out (padding) position: its hidden state is not "
                                      "computed by the packed engine")
        return packed_index[sel].to(torch.int32).contiguous(), mlm_labels[sel].to(torch.int32).contiguous()

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None,
                head_mask=None, inputs_embeds=None, labels=None, mlm_labels=None, output_attentions=None,
                output_hidden_states=None, doc_positions=None, return_mlm_scores=False):
        out = super().forward(input_ids=input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids,
                              position_ids=position_ids, head_mask=head_mask, inputs_embeds=inputs_embeds,
                              output_attentions=output_attentions, output_hidden_states=output_hidden_states,
                              doc_positions=doc_positions)
        logits = out[0]
        outputs = (logits,) if labels is not None else ((),)       # bert_model.py:167-172
        scores = None
        if mlm_labels is not None:
            lib, h = _lib.load(), self._ensure_engine()
            rows, labs = self._mlm_rows(mlm_labels.to(self.device),
                                        None if attention_mask is None else attention_mask.to(self.device))
            n = rows.numel()
            loss = torch.zeros(1, dtype=torch.float32, device=self.device)
            if return_mlm_scores:
                scores = torch.empty((n, self.config.vocab_size), dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(lib.clip_mlm_head(h, _lib.ptr(rows), _lib.ptr(labs), n, _lib.ptr(scores), _lib.ptr(loss), 0,
                                             _lib.stream_ptr()))
                if labels is not None:                              # + switch loss (bert_model.py:180-183)
                    lab = labels.to(self.device).reshape(-1).to(torch.int32).contiguous()
                    sw_logits = logits
                    if bool((lab == -100).any()):   # CrossEntropyLoss ignores -100 here too (bert_model.py:177-183)
                        keep = lab != -100
                        lab, sw_logits = lab[keep].contiguous(), logits[keep].contiguous()
                    if lab.numel() == 0:
                        loss += float("nan")        # torch: the mean over zero rows
                    else:
                        row_loss = torch.empty(lab.numel(), dtype=torch.float32, device=self.device)
                        _lib.check(lib.clip_ce_loss(_lib.ptr(sw_logits), _lib.ptr(lab), lab.numel(), self.num_labels,
                                                    _lib.ptr(row_loss), _lib.ptr(loss), 1, _lib.stream_ptr()))
            outputs = (loss[0],) + outputs
        return outputs + ((scores,) if return_mlm_scores else ())

    __call__ = forward


class BertForSequenceMultilabelClassification(_EngineModel):
    """ref: sentclf/bert_model.py:189-251 — BertModel pooler output -> Linear(H, num_labels)."""
    head_kind = 1

    def focus_index(self, ids, cu_seqlens, check=True, return_err=False):
        return self._focus(ids, cu_seqlens, -1, 0, check, return_err)


def args_to_model(n_context_sentences):
    """ref: sentclf/bert.py:361-366"""
    if n_context_sentences == 0:
        return BertForSequenceMultilabelClassification
    return BertSequenceMultilabelClassificationContext
