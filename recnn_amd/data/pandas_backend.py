"""Global pandas/modin switch used by the one-time CSV ETL (reference: recnn/data/pandas_backend.py).
Setup-only; not on the hot path."""


class PandasBackend:
    def __init__(self):
        self.backend = None
        self.type = "pandas"
        self.set()

    def set(self, backend="pandas"):
        if backend not in ("pandas", "modin"):
            print("Wrong backend specified! Usage: pd.set('pandas') or pd.set('modin'); using pandas")
            backend = "pandas"
        if backend == "modin":
            from modin import pandas as impl
        else:
            import pandas as impl
        self.type = backend
        self.backend = impl

    def get(self):
        return self.backend

    def get_type(self):
        return self.type


pd = PandasBackend()
