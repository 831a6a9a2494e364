"""`python -m time_series_spark_amd.scorer_driver config.yaml` -- the reference's
/root/reference/src/scorer_spark_driver.py:6-21 without Spark: load the YAML config
(keys io.models, io.forecasts, forecast.periods, forecast.frequency as in
/root/reference/config/example_scorer_app_config.yaml), forecast every stored model on the GPU in
one batched call, write the forecast CSV."""
import sys

import yaml

from .jobs.prophet_scorer import ProphetScorer


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) != 2:
        print("arg1 must be the config YAML")
        return 1
    with open(argv[1]) as file:
        config = yaml.safe_load(file)
    print(f"config: {config}")
    ProphetScorer.score(None, config)
    return 0


if __name__ == '__main__':
    sys.exit(main())
