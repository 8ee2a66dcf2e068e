"""Import the vLLM symbols the backend subclasses, or minimal stand-ins when vLLM is not installed
(this build container and the GPU test box have no vLLM; the stand-ins only make the module
importable and testable against the recorded interface, SURVEY.md §7.3 "vLLM API drift")."""

from __future__ import annotations

import enum
import logging

try:  # pragma: no cover - exercised only where vLLM is installed
    from vllm.attention.backends.abstract import AttentionBackend, AttentionImpl, AttentionMetadata, AttentionType
    from vllm.logger import init_logger
    from vllm.v1.attention.backends.utils import AttentionMetadataBuilder, CommonAttentionMetadata

    HAVE_VLLM = True
    try:
        from vllm.v1.attention.backends.utils import make_local_attention_virtual_batches
    except ImportError:
        make_local_attention_virtual_batches = None
except ImportError:
    HAVE_VLLM = False
    make_local_attention_virtual_batches = None

    class AttentionBackend:  # noqa: D101
        pass

    class AttentionImpl:  # noqa: D101
        pass

    class AttentionMetadata:  # noqa: D101
        pass

    class AttentionType(str, enum.Enum):  # noqa: D101
        DECODER = "decoder"
        ENCODER = "encoder"
        ENCODER_ONLY = "encoder_only"
        ENCODER_DECODER = "encoder_decoder"

    class _Generic:
        def __class_getitem__(cls, item):
            return cls

    class AttentionMetadataBuilder(_Generic):  # noqa: D101
        pass

    class CommonAttentionMetadata:  # noqa: D101
        pass

    def init_logger(name):
        logger = logging.getLogger(name)
        if not hasattr(logger, "warning_once"):
            seen = set()

            def warning_once(msg, *args):
                if msg not in seen:
                    seen.add(msg)
                    logger.warning(msg, *args)

            logger.warning_once = warning_once
        return logger
