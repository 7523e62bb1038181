"""Short, uniform kernel names for the profile tables.  rocprofv3's own demangler renders the bfloat16 template argument
(mangled DF16b) as garbage ("bool _Accum", or leaves the whole name mangled), so the profiling scripts ask for mangled
names (rocprofv3 -M) and this module demangles the subset the library uses: ssmi::<name><builtin types, integers, bools>.
Anything else (runtime copy kernels, torch kernels) is passed through with its argument list cut off."""
import re

_BUILTIN = {"f": "float", "d": "double", "i": "int", "j": "unsigned", "l": "long", "m": "unsigned long", "b": "bool",
            "h": "unsigned char", "t": "unsigned short", "DF16b": "bf16", "Dh": "half"}


def _template_args(s, i):
    """Parses I <arg>* E starting at s[i] == 'I'.  Returns (list of strings, index after the closing E) or (None, i)."""
    args = []
    i += 1
    while i < len(s) and s[i] != "E":
        if s.startswith("DF16b", i):
            args.append("bf16")
            i += 5
        elif s[i] == "L":                      # literal: L <type> <value> E
            m = re.match(r"L([a-zA-Z])(n?)(\d+)E", s[i:])
            if not m:
                return None, i
            v = ("-" if m.group(2) else "") + m.group(3)
            args.append({"0": "false", "1": "true"}.get(v, v) if m.group(1) == "b" else v)
            i += m.end()
        elif s[i] in _BUILTIN:
            args.append(_BUILTIN[s[i]])
            i += 1
        else:
            return None, i
    return args, i + 1


def demangle(name):
    """ssmi kernels: 'conv_fwd_kernel<float, 1, true, 0>'; other names: unchanged."""
    m = re.match(r"_ZN4ssmi(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    start = m.end()
    base = name[start:start + n]
    i = start + n
    if i < len(name) and name[i] == "I":
        args, _ = _template_args(name, i)
        if args is None:
            return base + "<?>"
        return "{}<{}>".format(base, ", ".join(args))
    return base


def short(name, strip_type=False):
    """Kernel name without return type, namespace and argument list; strip_type drops a leading storage-type template
    argument (the library's own per-variant records name variants without it)."""
    name = name.strip().strip('"')
    if name.startswith("_Z"):
        name = demangle(name)
    name = re.sub(r"^void\s+", "", name).replace("ssmi::", "").replace("__bf16", "bf16")
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    name = name[:cut].strip()
    if strip_type:
        name = re.sub(r"<(float|bf16), ", "<", name)
        name = re.sub(r"<(float|bf16)>$", "", name)
    return name


def storage_type(name):
    m = re.search(r"<(float|bf16)[,>]", short(name))
    return m.group(1) if m else None


if __name__ == "__main__":
    import sys
    for a in sys.argv[1:]:
        print(short(a))
