"""Short, stable names for the kernels rocprofv3 reports (template arguments folded into a precision tag)."""
import re

# kernels with an instrumented (visit-counting) instantiation: bench.py runs it for ONE untimed frame; it is listed apart
COUNTED = ("k_trace_gi", "k_trace_di", "k_fill_gbuffer", "k_trace_rays")


def count_flag(kernel, bools):
    """COUNT is the last bool template argument, except in k_trace_di<T, FIRST, COMPOSE, COUNT, WIDE>."""
    return bools[2] if kernel == "k_trace_di" and len(bools) >= 4 else bools[-1]


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN4lprt(\d+)", name)  # rocprofv3 leaves the _Float16 instantiations mangled
    if m:
        n = int(m.group(1))
        k = name[m.end():m.end() + n]
        targs = name[m.end() + n:]
        variant = ""
        if k in ("fastslab", "refslab"):  # the ray-casting kernels live in a namespace per slab variant (csrc/kernels.hpp)
            variant = "{refslab}" if k == "refslab" else ""
            m2 = re.match(r"(\d+)", targs)
            n = int(m2.group(1))
            k = targs[m2.end():m2.end() + n]
            targs = targs[m2.end() + n:]
        prec = "f16" if targs.startswith("IDF16_") else "f32"
        head = targs.split("EEv")[0]
        ints = re.findall(r"Li(\d+)E", head)
        bools = re.findall(r"Lb([01])E", head)
        count = "+count" if k in COUNTED and bools and count_flag(k, bools) == "1" else ""
        return f"{k}{count}{'<' + ','.join(ints) + '>' if ints else ''}[{prec}]{variant}"
    m = re.match(r"(?:lprt::)?(?:fastslab::|refslab::)?(k_\w+)<(.*)>\(", name)
    if not m:
        return name.split("(")[0][:60]
    k, targs = m.group(1), m.group(2)
    prec = "f16" if "_Float16" in targs or "half" in targs else "f32"
    extra = [a.strip() for a in targs.split(",") if a.strip().isdigit()]
    bools = [a.strip() for a in targs.split(",") if a.strip() in ("true", "false")]
    count = "+count" if k in COUNTED and bools and count_flag(k, bools) == "true" else ""
    return f"{k}{count}{'<' + ','.join(extra) + '>' if extra else ''}[{prec}]"


