set -x
cd $GRAFT_REPO_ROOT
( time python3 run_me.py icrl -er 10 -tk 0.01 -cl 20 -bi 10 -ft 2e5 -ni 30 -tei HCWithPos-v0 -eei HCWithPosTest-v0 -clr 0.05 -aclr 0.9 -crc 0.5 -psis -ctkno 2.5 -nt 64 -s 0 -v 1 2>&1 | grep -v amdgpu | tail -2 | cut -c1-700 ) 2>&1 | tail -6
( time python3 run_me.py icrl -er 10 -ep tests/golden/expert_ant.npz -tk 0.01 -cl 40 40 -bi 10 -ft 2e5 -ni 4 -tei AntWall-v0 -eei AntWallTest-v0 -clr 0.005 -crc 0.6 -psis -ctkno 2.5 -nt 256 --n_steps 512 -bs 128 -s 0 -v 1 2>&1 | grep -v amdgpu | tail -1 | cut -c1-600 ) 2>&1 | tail -5
( time python3 run_me.py cpg -tei AntWallBroken-v0 -eei AntWallBrokenTest-v0 -cp tests/golden/cn_antbroken.npz -t 1e6 -nt 512 --n_steps 256 -bs 128 -s 0 2>&1 | grep -v amdgpu | tail -2 | cut -c1-500 ) 2>&1 | tail -6
( time python3 run_me.py icrl -er 20 -ep tests/golden/expert_lgw.npz -tei LGW-v0 -eei CLGW-v0 -tk 0.01 -cl 20 -clr 0.003 -ft 0.5e5 -ni 10 -bi 20 -dno -dnr -dnc -s 0 -v 1 2>&1 | grep -v amdgpu | tail -1 | cut -c1-500 ) 2>&1 | tail -5
