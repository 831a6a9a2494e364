"""`python -m time_series_spark_amd.modeler_driver config.yaml` -- the reference's
/root/reference/src/modeler_spark_driver.py:6-22 without Spark: load the YAML config
(keys io.input, io.models, model.floor, model.cap_multiplier as in
/root/reference/config/example_modeler_app_config.yaml), fit every (series_id, dim_id) of the
input directory on the GPU in one batched call, write the model parquet."""
import sys

import yaml

from .jobs.prophet_modeler import ProphetModeler


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) != 2:
        print("arg1 must be the config YAML")
        return 1
    with open(argv[1]) as file:
        config = yaml.safe_load(file)
    print(f"config: {config}")
    ProphetModeler.model(None, config, return_frame=False)
    return 0


if __name__ == '__main__':
    sys.exit(main())
