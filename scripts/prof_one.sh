# rocprofv3 kernel stats of one script: bash scripts/prof_one.sh <script.py> [args]  (prints the top kernels)
export TMPDIR=/tmp
d=/tmp/prof_one_$$
(cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o r -- python3 $GRAFT_REPO_ROOT/$1 ${@:2} > /dev/null 2>&1)
python3 $GRAFT_REPO_ROOT/scripts/prof_summary.py $d/r_results.db 2>/dev/null | head -${TOPN:-12} | python3 -c "
import sys,re
for l in sys.stdin:
    m=re.search(r'calls=.*', l); name=re.sub(r'\(.*','',l.split('calls=')[0]).strip()
    print(name[:60].ljust(62), m.group(0) if m else '')"
rm -rf $d
