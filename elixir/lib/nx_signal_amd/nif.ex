defmodule NxSignalAMD.NIF do
  @moduledoc false
  # Thin loader of nif/nxsig_nif.c.  Every function is replaced at load time; the bodies below only run when
  # the shared object is missing.
  @on_load :load_nif

  def load_nif do
    path = :filename.join(:code.priv_dir(:nx_signal_amd), ~c"nxsig_nif")
    :erlang.load_nif(path, 0)
  end

  def ctx_create(_device), do: :erlang.nif_error(:nif_not_loaded)
  def window(_kind, _n, _periodic, _beta, _eps), do: :erlang.nif_error(:nif_not_loaded)
  def firwin(_taps, _cutoff, _kind, _beta, _pass_zero, _scale, _fs), do: :erlang.nif_error(:nif_not_loaded)
  def stft(_ctx, _x, _length, _batch, _window, _params), do: :erlang.nif_error(:nif_not_loaded)
  def istft(_ctx, _z, _frames, _batch, _window, _params), do: :erlang.nif_error(:nif_not_loaded)
  def fir(_ctx, _x, _length, _batch, _taps, _mode), do: :erlang.nif_error(:nif_not_loaded)
  def to_device(_ctx, _bin), do: :erlang.nif_error(:nif_not_loaded)
  def from_device(_buf), do: :erlang.nif_error(:nif_not_loaded)
  def stft_dev(_ctx, _x, _length, _batch, _window, _params), do: :erlang.nif_error(:nif_not_loaded)
  def istft_dev(_ctx, _z, _frames, _batch, _window, _params), do: :erlang.nif_error(:nif_not_loaded)
  def spectrum_mul_dev(_ctx, _z, _rows, _fft_length, _h), do: :erlang.nif_error(:nif_not_loaded)
end
