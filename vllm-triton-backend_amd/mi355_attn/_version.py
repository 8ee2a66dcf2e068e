"""ONE version source: `MI355_ATTN_VERSION` of include/mi355_attn.h (major*10000 + minor*100 + patch). The built library
reports the same number through `mi355_attn_version()`; `tests/test_cpu_host.py` checks that the three agree."""

import os
import re

_HEADER_CANDIDATES = (
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "include", "mi355_attn.h"),   # the repository layout
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "mi355_attn.h"),                           # an installed wheel ships a copy
)


def version_number() -> int:
    for path in _HEADER_CANDIDATES:
        try:
            m = re.search(r"#define\s+MI355_ATTN_VERSION\s+(\d+)", open(path).read())
        except OSError:
            continue
        if m:
            return int(m.group(1))
    raise RuntimeError("include/mi355_attn.h (MI355_ATTN_VERSION) not found beside the package")


def version_string(n: int | None = None) -> str:
    n = version_number() if n is None else n
    return f"{n // 10000}.{n // 100 % 100}.{n % 100}"
