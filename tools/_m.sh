run() { echo "== $CFG $*"; env "$@" LCMI_PTS=0.01 LCMI_FU=10 python3 tools/joint_speed.py $CFG 2>&1 | grep -E "us/iter|^loss"; }
CFG="125 128 4 150"; run A=1; run A=1
CFG="200 128 4 100"; run A=1
CFG="1000 128 4 30"; run A=1
