"""vLLM V1 attention backend on libmi355_attn.so.

Mirrors the reference's TritonAttentionBackend / TritonAttentionImpl / TritonAttentionMetadataBuilder /
TritonAttentionMetadata (LIB/backend/triton_attn.py:60-470): same static contract, same constructor
and forward signatures, same error behaviour. forward() calls two registered ops, torch.ops.mi355_attn.
reshape_and_cache_flash and .unified_attention (mi355_attn/ops.py): two C-ABI calls on the current stream.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, ClassVar, Optional

import torch

from .. import _lib
from .. import ops as _ops  # noqa: F401  (registers torch.ops.mi355_attn.*)
from ._vllm_shim import (
    AttentionBackend,
    AttentionImpl,
    AttentionMetadataBuilder,
    AttentionType,
    CommonAttentionMetadata,
    init_logger,
    make_local_attention_virtual_batches,
)

import os  # noqa: E402

logger = init_logger(__name__)
_FUSED_DECODE_WRITE = os.environ.get("MI355_FUSED_DECODE_WRITE", "1") != "0"   # A/B switch: 0 = always two launches
_FUSED_PREFILL_WRITE = os.environ.get("MI355_FUSED_PREFILL_WRITE", "1") != "0"   # A/B switch: 0 = cache write and attention as two calls for prefill steps too
_lib.load()  # fail at import, not at the first forward, if the HIP library is missing


@dataclass
class MI355AttentionMetadata:
    """Per-step metadata (reference: TritonAttentionMetadata, triton_attn.py:60-103)."""

    num_actual_tokens: int  # Number of tokens excluding padding.
    max_query_len: int
    avg_query_len: int
    avg_seq_len: int
    query_start_loc: torch.Tensor
    max_seq_len: int
    seq_lens: torch.Tensor
    block_table: torch.Tensor
    slot_mapping: torch.Tensor

    # cascade attention is never used (use_cascade_attention -> False), kept for interface parity
    use_cascade: bool
    common_prefix_len: int
    cu_prefix_query_lens: Optional[torch.Tensor]
    prefix_kv_lens: Optional[torch.Tensor]
    suffix_kv_lens: Optional[torch.Tensor]

    scheduler_metadata: Optional[torch.Tensor] = None
    prefix_scheduler_metadata: Optional[torch.Tensor] = None
    # (not in the reference) query tokens of a decode row when speculative decoding is on, 1 + num_speculative_tokens: lets a
    # step that mixes prefills with verification rows send the latter to the decode launch (0: the library's own choice)
    decode_rows_hint: int = 0

    @dataclass
    class LocalAttentionMetadata:
        local_query_start_loc: torch.Tensor
        local_seqused_k: torch.Tensor
        local_block_table: torch.Tensor
        local_max_query_len: int
        local_max_seq_len: int
        local_avg_query_len: int
        local_avg_seq_len: int
        local_scheduler_metadata: Optional[torch.Tensor]

    local_attn_metadata: Optional[LocalAttentionMetadata] = None


class MI355AttentionMetadataBuilder(AttentionMetadataBuilder[MI355AttentionMetadata]):
    """Host-side, once per scheduler step (reference: triton_attn.py:106-233)."""

    full_cudagraph_supported: ClassVar[bool] = True

    def __init__(self, runner, kv_cache_spec, block_table):
        self.runner = runner
        self.block_size = kv_cache_spec.block_size
        self.kv_cache_spec = kv_cache_spec
        self.block_table = block_table
        # host-known and fixed for the engine's life (capture-stable)
        spec = getattr(getattr(runner, "vllm_config", None), "speculative_config", None) or getattr(runner, "speculative_config", None)
        self.decode_rows_hint = 1 + int(getattr(spec, "num_speculative_tokens", 0) or 0) if spec is not None else 0

    def build_for_cudagraph_capture(self, common_attn_metadata: CommonAttentionMetadata) -> MI355AttentionMetadata:
        attn_metadata = self.build(0, common_attn_metadata)
        # capture with seq_lens = 1 so that the captured kernels do minimal work (triton_attn.py:124-127);
        # the kernels clamp negative context lengths, so this is harmless for query_len > 1 rows
        attn_metadata.seq_lens.fill_(1)
        return attn_metadata

    # ---- the three host-side pieces of a step's metadata ---------------------------------------------------------
    def _length_stats(self, n_seqs: int):
        """(max, mean) of the step's key lengths and the mean query length, from the runner's numpy mirrors: host
        integers only, the kernels' split plans are functions of them (no device read-back)."""
        lens = self.runner.seq_lens_np[:n_seqs]
        return int(lens.max()), int(lens.mean()), int(self.runner.query_start_loc_np[n_seqs] / n_seqs)

    def _stage_slot_mapping(self, n_tokens: int) -> torch.Tensor:
        """This step's slots to the device; every row behind them becomes -1, which the cache write - separate or fused
        into the decode launch - skips (a captured graph is replayed with padded rows; reference: triton_attn.py:149-151)."""
        table = self.block_table
        table.slot_mapping[:n_tokens].copy_(table.slot_mapping_cpu[:n_tokens], non_blocking=True)
        table.slot_mapping[n_tokens:].fill_(-1)
        return table.slot_mapping[:n_tokens]

    def _local_window_metadata(self, n_seqs: int, pages: torch.Tensor):
        """Chunked-local (iRoPE) layers attend inside fixed windows: vLLM's helper rewrites the batch into one virtual
        sequence per (request, window), and the kernels run on that batch unchanged (reference: triton_attn.py:157-190)."""
        chunk = getattr(self.runner, "attention_chunk_size", None)
        if chunk is None:
            return None
        if make_local_attention_virtual_batches is None:
            raise NotImplementedError("local (chunked) attention needs vLLM's make_local_attention_virtual_batches")
        q_lens, q_starts, k_lens, virtual_pages = make_local_attention_virtual_batches(
            chunk, self.runner.query_start_loc_np[: n_seqs + 1], self.runner.seq_lens_np[:n_seqs], pages, self.block_size)
        count = max(len(k_lens), 1)
        to_dev = lambda a: torch.from_numpy(a).to(self.runner.device, non_blocking=True)  # noqa: E731
        return MI355AttentionMetadata.LocalAttentionMetadata(
            local_query_start_loc=to_dev(q_starts), local_seqused_k=to_dev(k_lens), local_block_table=virtual_pages,
            local_max_query_len=int(q_lens.max()), local_max_seq_len=int(k_lens.max()),
            local_avg_query_len=int(q_lens.sum() / count), local_avg_seq_len=int(k_lens.sum() / count),
            local_scheduler_metadata=None)

    def build(self, common_prefix_len: int, common_attn_metadata: CommonAttentionMetadata) -> MI355AttentionMetadata:
        """One step's metadata (reference: TritonAttentionMetadataBuilder.build, triton_attn.py:130-227). Cascade attention
        is off for this backend (use_cascade_attention -> False, :279-281): vLLM never passes a common prefix and the
        cascade tensors stay None."""
        cm = common_attn_metadata
        n_seqs, n_tokens = cm.num_reqs, cm.num_actual_tokens
        longest, mean_keys, mean_queries = self._length_stats(n_seqs)
        pages = self.block_table.get_device_tensor()[:n_seqs]
        return MI355AttentionMetadata(
            num_actual_tokens=n_tokens, max_query_len=cm.max_query_len, query_start_loc=cm.query_start_loc,
            max_seq_len=longest, seq_lens=cm.seq_lens, block_table=pages, slot_mapping=self._stage_slot_mapping(n_tokens),
            use_cascade=False, common_prefix_len=common_prefix_len,
            cu_prefix_query_lens=None, prefix_kv_lens=None, suffix_kv_lens=None,
            local_attn_metadata=self._local_window_metadata(n_seqs, pages), prefix_scheduler_metadata=None,
            avg_query_len=mean_queries, avg_seq_len=mean_keys, decode_rows_hint=self.decode_rows_hint)

    def can_run_in_cudagraph(self, common_attn_metadata: CommonAttentionMetadata) -> bool:
        return True  # static launch grids, caller-owned workspace, no host sync


class MI355AttentionBackend(AttentionBackend):
    """Static contract (reference: TritonAttentionBackend, triton_attn.py:236-285)."""

    accept_output_buffer: bool = True

    @classmethod
    def get_supported_head_sizes(cls) -> list[int]:
        return [32, 64, 96, 128, 160, 192, 224, 256]

    @classmethod
    def validate_head_size(cls, head_size: int) -> None:
        supported_head_sizes = cls.get_supported_head_sizes()
        if head_size not in supported_head_sizes:
            attn_type = cls.__name__.removesuffix("Backend")
            raise ValueError(
                f"Head size {head_size} is not supported by {attn_type}. "
                f"Supported head sizes are: {supported_head_sizes}. "
                "Set VLLM_ATTENTION_BACKEND=FLEX_ATTENTION to use "
                "FlexAttention backend which supports all head sizes."
            )

    @staticmethod
    def get_name() -> str:
        # the reference's name, so VLLM_ATTENTION_BACKEND=TRITON_ATTN_VLLM_V1 scripts keep working
        return "TRITON_ATTN_VLLM_V1"

    @staticmethod
    def get_impl_cls() -> type["MI355AttentionImpl"]:
        return MI355AttentionImpl

    @staticmethod
    def get_metadata_cls() -> type["MI355AttentionMetadata"]:
        return MI355AttentionMetadata

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int, head_size: int) -> tuple[int, ...]:
        if block_size % 16 != 0:
            raise ValueError("Block size must be a multiple of 16.")
        return (2, num_blocks, block_size, num_kv_heads, head_size)

    @staticmethod
    def use_cascade_attention(*args, **kwargs) -> bool:
        return False

    @staticmethod
    def get_builder_cls() -> type["MI355AttentionMetadataBuilder"]:
        return MI355AttentionMetadataBuilder


class MI355AttentionImpl(AttentionImpl):
    """Per-layer forward (reference: TritonAttentionImpl, triton_attn.py:288-470)."""

    def __init__(
        self,
        num_heads: int,
        head_size: int,
        scale: float,
        num_kv_heads: int,
        alibi_slopes: Optional[list[float]],
        sliding_window: Optional[int],
        kv_cache_dtype: str,
        blocksparse_params: Optional[dict[str, Any]] = None,
        logits_soft_cap: Optional[float] = None,
        attn_type: AttentionType = AttentionType.DECODER,
        kv_sharing_target_layer_name: Optional[int] = None,
        use_irope: bool = False,
    ) -> None:
        if blocksparse_params is not None:
            raise ValueError("MI355Attention does not support block-sparse attention.")
        self.num_heads = num_heads
        self.head_size = head_size
        self.scale = float(scale)
        self.num_kv_heads = num_kv_heads
        if alibi_slopes is not None:
            alibi_slopes = torch.tensor(alibi_slopes, dtype=torch.float32)
        self.alibi_slopes = alibi_slopes
        self._alibi_dev: Optional[torch.Tensor] = None
        if sliding_window is None:
            self.sliding_window = (-1, -1)
        else:
            self.sliding_window = (sliding_window - 1, 0)
        self.kv_cache_dtype = kv_cache_dtype
        if logits_soft_cap is None:
            logits_soft_cap = 0  # 0 means no soft cap
        self.logits_soft_cap = logits_soft_cap
        self.kv_sharing_target_layer_name = kv_sharing_target_layer_name
        self.use_irope = use_irope
        self.num_queries_per_kv = self.num_heads // self.num_kv_heads
        self._q_scale_checked = False

        MI355AttentionBackend.validate_head_size(head_size)

        if attn_type != AttentionType.DECODER:
            raise NotImplementedError(
                "Encoder self-attention and encoder/decoder cross-attention are not implemented for MI355AttentionImpl"
            )
        # gfx950 speaks OCP fp8 (e4m3fn), unlike MI300's fnuz
        self.fp8_dtype = torch.float8_e5m2 if kv_cache_dtype == "fp8_e5m2" else torch.float8_e4m3fn
        logger.warning_once("Using mi355-attn attention PLUGIN V1 (hand-written gfx950 HIP kernels).")

    def _check_q_scale(self, layer) -> None:
        """The reference asserts `layer._q_scale == 1.0` on every forward (triton_attn.py:412): with a device tensor that
        is a host sync per layer and step, and an error under graph capture. Checked ONCE per layer here (the scale is a
        checkpoint constant), from the host-side float when the layer carries one, never while a stream is capturing."""
        if self._q_scale_checked:
            return
        qs = getattr(layer, "_q_scale_float", None)
        if qs is None:
            qs = getattr(layer, "_q_scale", 1.0)
            if isinstance(qs, torch.Tensor):
                if qs.is_cuda and torch.cuda.is_current_stream_capturing():
                    return                      # not now; the warm-up run before the capture has normally checked already
                qs = float(qs)
        assert qs == 1.0, "A non 1.0 q_scale is not currently supported."
        self._q_scale_checked = True

    def forward(
        self,
        layer: torch.nn.Module,
        query: torch.Tensor,
        key: torch.Tensor,
        value: torch.Tensor,
        kv_cache: torch.Tensor,
        attn_metadata: MI355AttentionMetadata,
        output: Optional[torch.Tensor] = None,
        output_scale: Optional[torch.Tensor] = None,
    ) -> torch.Tensor:
        """query [T, Hq, D]; key/value [T, Hk, D]; kv_cache [2, num_blocks, block_size, Hk, D];
        output [T, Hq, D] (or [T, Hq*D]) is written in place and returned."""
        assert output is not None, "Output tensor must be provided."
        if output_scale is not None:
            raise NotImplementedError("fused output quantization is not yet supported for MI355AttentionImpl")
        if attn_metadata is None:
            return output  # profiling run
        assert attn_metadata.use_cascade is False

        num_actual_tokens = attn_metadata.num_actual_tokens
        key_cache, value_cache = kv_cache.unbind(0)

        # A decode step (every sequence one query token, no window / soft-cap / ALiBi / local attention): ONE launch - the
        # wave that owns a sequence's last tile stores the new token's K/V row into its page and attends over it
        # (the op falls back to the two calls below when the fused kernel does not serve the shape).
        # vLLM's slot_mapping of such a step points at position seq_len - 1 of each sequence, which is where the kernel writes.
        plain = self.alibi_slopes is None and self.sliding_window == (-1, -1) and not self.logits_soft_cap
        if (attn_metadata.max_query_len == 1 and plain and self.kv_sharing_target_layer_name is None
                and not (self.use_irope and attn_metadata.local_attn_metadata is not None) and _FUSED_DECODE_WRITE):
            q = query[:num_actual_tokens]
            out = output[:num_actual_tokens]
            if out.dim() == 2:
                out = out.view(-1, self.num_heads, self.head_size)
            if q.dim() == 2:
                q = q.view(-1, self.num_heads, self.head_size)
            if self.kv_cache_dtype.startswith("fp8"):
                self._check_q_scale(layer)
            torch.ops.mi355_attn.decode_attention_and_cache_write(
                q, key, value, key_cache, value_cache, out, attn_metadata.query_start_loc, attn_metadata.seq_lens, int(attn_metadata.max_seq_len),
                float(self.scale), attn_metadata.block_table, attn_metadata.slot_mapping, layer._k_scale, layer._v_scale, self.kv_cache_dtype)
            return output

        # A plain step that is not a decode step (round 4): the op issues ONE launch where the short-prompt kernel serves the
        # step with the cache write inside (library 0.6.0: a prompt of a few hundred tokens - the reference's own latency
        # protocol), and the pair of calls below otherwise - same cache bytes, same output. MI355_FUSED_PREFILL_WRITE=0 keeps the pair.
        if (plain and attn_metadata.max_query_len > 1 and self.kv_sharing_target_layer_name is None and not self.kv_cache_dtype.startswith("fp8")
                and not (self.use_irope and attn_metadata.local_attn_metadata is not None) and _FUSED_PREFILL_WRITE):
            q = query[:num_actual_tokens]
            out = output[:num_actual_tokens]
            if out.dim() == 2:
                out = out.view(-1, self.num_heads, self.head_size)
            if q.dim() == 2:
                q = q.view(-1, self.num_heads, self.head_size)
            torch.ops.mi355_attn.prefill_attention_and_cache_write(
                q, key, value, key_cache, value_cache, out, attn_metadata.query_start_loc, int(attn_metadata.max_query_len), attn_metadata.seq_lens,
                int(attn_metadata.max_seq_len), float(self.scale), attn_metadata.block_table, attn_metadata.slot_mapping, layer._k_scale, layer._v_scale,
                self.kv_cache_dtype, int(getattr(attn_metadata, "decode_rows_hint", 0)))
            return output

        if self.kv_sharing_target_layer_name is None:
            torch.ops.mi355_attn.reshape_and_cache_flash(key, value, key_cache, value_cache, attn_metadata.slot_mapping, self.kv_cache_dtype,
                                                         layer._k_scale, layer._v_scale)

        if self.kv_cache_dtype.startswith("fp8"):
            self._check_q_scale(layer)
            # Q stays in its own dtype: K/V are dequantised in the kernel (the reference skips Q
            # quantisation on ROCm as well, triton_attn.py:414-420); the op views the uint8 cache as gfx950's OCP fp8

        use_local_attn = self.use_irope and attn_metadata.local_attn_metadata is not None
        if use_local_attn:
            lm = attn_metadata.local_attn_metadata
            cu_seqlens_q, seqused_k = lm.local_query_start_loc, lm.local_seqused_k
            max_seqlen_q, max_seqlen_k = lm.local_max_query_len, lm.local_max_seq_len
            block_table = lm.local_block_table
        else:
            cu_seqlens_q, seqused_k = attn_metadata.query_start_loc, attn_metadata.seq_lens
            max_seqlen_q, max_seqlen_k = attn_metadata.max_query_len, attn_metadata.max_seq_len
            block_table = attn_metadata.block_table

        if self.alibi_slopes is not None and (self._alibi_dev is None or self._alibi_dev.device != query.device):
            self._alibi_dev = self.alibi_slopes.to(query.device)

        q = query[:num_actual_tokens]
        out = output[:num_actual_tokens]
        if out.dim() == 2:
            out = out.view(-1, self.num_heads, self.head_size)
        if q.dim() == 2:
            q = q.view(-1, self.num_heads, self.head_size)

        # (avg_seqlen_q/k only fed the reference's autotuner keys, triton_unified_attention.py:878-881: not passed on)
        torch.ops.mi355_attn.unified_attention(
            q, key_cache, value_cache, out, cu_seqlens_q, int(max_seqlen_q), seqused_k, int(max_seqlen_k), float(self.scale),
            int(self.sliding_window[0]), int(self.sliding_window[1]), block_table, float(self.logits_soft_cap), layer._k_scale, layer._v_scale,
            self._alibi_dev, self.kv_cache_dtype, int(getattr(attn_metadata, "decode_rows_hint", 0)))
        return output
