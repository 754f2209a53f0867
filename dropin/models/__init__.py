"""Put this directory on PYTHONPATH in front of the reference checkout: train.py then imports
the MI355X `models` package unchanged."""
