"""static instruction mix of one kernel in a hipcc -S output:  isa_mix.py file.s kernel_substring"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(sys.argv[2]), l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
ops = []
for l in lines[start + 1:end]:
    l = l.split(";")[0].strip()
    if not l or l.endswith(":") or l.startswith("."):
        continue
    ops.append(l.split()[0])
c = collections.Counter()
for o in ops:
    if o.startswith("v_mfma"): c["mfma"] += 1
    elif o.startswith(("global_load", "buffer_load", "flat_load")): c["vmem_load"] += 1
    elif o.startswith(("global_store", "buffer_store")): c["vmem_store"] += 1
    elif o.startswith("global_atomic"): c["vmem_atomic"] += 1
    elif o.startswith("ds_"): c["lds"] += 1
    elif o.startswith("v_"): c["valu"] += 1
    elif o.startswith("s_waitcnt"): c["s_waitcnt"] += 1
    elif o.startswith(("s_cbranch", "s_branch")): c["branch"] += 1
    elif o.startswith("s_"): c["salu"] += 1
    else: c[o] += 1
print("total", len(ops), dict(c))
print("top valu:", collections.Counter(o for o in ops if o.startswith("v_")).most_common(14))
print("lds:", collections.Counter(o for o in ops if o.startswith("ds_")).most_common(8))
