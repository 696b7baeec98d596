#!/bin/bash
# Resource usage of the gfx950 kernels in a compiled object / shared library, from the code object's notes:
#   scripts/dev/kres.sh libwave_amd/csrc/wm_nn.o [pattern]
# columns: vgprs (arch) agprs sgprs vgpr-spills sgpr-spills scratch-bytes/lane lds-bytes name
f=$1; pat=${2:-.}
tmp=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$f" "$tmp/fat.bin" 2>/dev/null
tgt=$(/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input="$tmp/fat.bin" 2>/dev/null | grep gfx950 | head -1)
[ -n "$tgt" ] && /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets="$tgt" --input="$tmp/fat.bin" --output="$tmp/dev.co" --unbundle 2>/dev/null
[ -s "$tmp/dev.co" ] || { echo "no gfx950 code object in $f"; exit 1; }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$tmp/dev.co" | python3 -c '
import sys, re
txt = sys.stdin.read()
pat = re.compile(sys.argv[1])
for blk in txt.split("  - .agpr_count:")[1:]:
    def g(k):
        m = re.search(r"\.%s:\s+(\S+)" % k, blk)
        return m.group(1) if m else "?"
    name = g("name")
    if not pat.search(name): continue
    import subprocess
    try: dn = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception: dn = name
    agpr = blk.split("\n")[0].strip()
    print("v%-4s a%-4s s%-4s vspill %-4s sspill %-4s scratch %-5s lds %-6s %s" % (g("vgpr_count"), agpr, g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), dn[:150]))
' "$pat"
rm -rf "$tmp"
