for sz in "360 540" "720 540" "720 1080" "1440 1080"; do
  echo "size $sz: $(PROF_MODES=adjust timeout 100 python scripts/prof_continuity.py $sz 75 2>&1 | grep '^lds' | sed 's/lds //')"
done
