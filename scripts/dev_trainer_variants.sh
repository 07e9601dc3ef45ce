cd /tmp && export TMPDIR=/tmp
for v in ""; do
  if [ -z "$v" ]; then unset ECGB_SO; else export ECGB_SO=libecgbyte_hip_$v.so; fi
  rm -rf /tmp/tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 /root/repo/scripts/dev_trainer_prof.py > /dev/null 2>&1
  echo "== variant ${v:-current}"; python3 /root/repo/scripts/dev_trainer_trace.py /tmp/tr
done
