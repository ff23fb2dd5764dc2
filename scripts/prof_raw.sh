# rocprofv3 kernel stats of one script, raw kernel names: bash scripts/prof_raw.sh <out.txt> <script.py> [args]
export TMPDIR=/tmp
d=/tmp/prof_raw_$$
(cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o r -- python3 $GRAFT_REPO_ROOT/$2 ${@:3} > /dev/null 2>&1)
python3 $GRAFT_REPO_ROOT/scripts/prof_summary.py $d/r_results.db 400 > $1 2>&1
rm -rf $d
