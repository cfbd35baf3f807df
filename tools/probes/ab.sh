#!/bin/bash
# usage: tools/probes/ab.sh "<M list>" <form> "<variant name>=<-D flags>" ...   (builds each probe variant, runs them in ONE gpurun call)
cd /root/repo
MS="$1"; FORM="$2"; shift 2
NAMES=""
for v in "$@"; do
  name="${v%%=*}"; flags="${v#*=}"; [ "$flags" = "$v" ] && flags=""
  SUFFIX="_$name" tools/probes/build_g4w.sh $flags 2>&1 | grep -v "^-rwx" | head -5
  NAMES="$NAMES $name"
done
/usr/local/graft/bin/gpurun --timeout 1200 -- "for s in $NAMES; do for m in $MS; do echo variant \$s; tools/probes/_bin/g4w_probe_\$s \$m $FORM | cut -c1-230; done; done > gpurun_out/g4w_probe.txt 2>&1" 2>&1 | tail -1
cat /root/repo/gpurun_out/g4w_probe.txt
