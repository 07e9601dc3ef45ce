# Dev-only: per-kernel totals and per-merge durations along the trainer's run, for every form in FORMS (ecgb_set_bpe_train_form) and corpus in CORPORA ("c2" and/or "walk").
cd /tmp && export TMPDIR=/tmp
for c in ${CORPORA:-c2}; do
for f in ${FORMS_LIST:-0 2}; do
  if [ "$c" = c2 ]; then export CORPUS=c2; else unset CORPUS; fi
  export FORMS=$f
  rm -rf /tmp/tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 /root/repo/scripts/dev_trainer_prof.py > /dev/null 2>&1
  echo "== corpus $c form $f"; python3 /root/repo/scripts/dev_trainer_trace.py /tmp/tr
done
done
