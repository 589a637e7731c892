cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in base hnoslp hlate; do
  if [ $v = base ]; then L=""; else L=icrl_amd/lib/var/libicrl_$v.so; fi
  echo -n "$v: "; ICRL_LIB=$L VARIANTS=auto,auto,auto python tools/train_only.py 2>&1 | grep "us/step" | awk '{printf "%s ", $4}'; echo
done; done
