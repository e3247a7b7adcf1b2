defmodule NxSignalAMD.Filters do
  @moduledoc """
  `NxSignal.Filters.firwin/3` (lib/nx_signal/filters.ex:147-279) and the streaming `fir/3` the reference lacks:
  `fir(x, taps, mode: :same)` == `NxSignal.Convolution.convolve(x, taps, method: :fft, mode: :same)` to fp32
  rounding, computed by overlap-save block FFT convolution on the GPU.
  """
  alias NxSignalAMD.NIF

  @windows %{hamming: 4, hann: 5, blackman: 3, bartlett: 1, rectangular: 0}
  @modes %{full: 0, same: 1, valid: 2}

  def firwin(num_taps, cutoff, opts \\ []) do
    opts = Keyword.validate!(opts, window: :hamming, pass_zero: true, scale: true, sampling_rate: 2.0, type: {:f, 32})

    if not is_list(cutoff) do
      raise ArgumentError, "cutoff must be a list of frequencies, got: #{inspect(cutoff)}"
    end

    {kind, beta} =
      case opts[:window] do
        {:kaiser, beta} -> {6, beta * 1.0}
        w when is_map_key(@windows, w) -> {@windows[w], 0.0}
        w -> raise ArgumentError, "unknown window #{inspect(w)}, supported: :hamming, :hann, :blackman, :bartlett, :rectangular, {:kaiser, beta}"
      end

    args = [num_taps, Enum.map(cutoff, &(&1 * 1.0)), kind, beta, b(opts[:pass_zero]), b(opts[:scale]), opts[:sampling_rate] * 1.0]

    case Nx.Type.normalize!(opts[:type]) do
      {:f, 64} ->
        {:ok, bin} = apply(NIF, :firwin_f64, args) |> NxSignalAMD.unwrap!()
        Nx.from_binary(bin, :f64)

      {:f, 32} ->
        {:ok, bin} = apply(NIF, :firwin, args) |> NxSignalAMD.unwrap!()
        Nx.from_binary(bin, :f32)

      other ->
        raise ArgumentError, "firwin: type must be {:f, 32} or {:f, 64}, got: #{inspect(other)}"
    end
  end

  def fir(x, taps, opts \\ [])

  # device-resident stream: filtered in HBM, the result stays there
  def fir(%NxSignalAMD.DeviceTensor{type: {:f, 32}} = x, taps, opts) do
    opts = Keyword.validate!(opts, mode: :same)

    if not is_map_key(@modes, opts[:mode]) do
      raise ArgumentError, "expected mode to be one of [:full, :same, :valid], got: #{inspect(opts[:mode])}"
    end

    r = tuple_size(x.shape)
    length = elem(x.shape, r - 1)
    batch_shape = Tuple.delete_at(x.shape, r - 1)
    hb = taps |> Nx.as_type(:f32) |> Nx.to_binary()

    {:ok, yref, n_out} =
      NIF.fir_dev(x.ctx, x.ref, length, Tuple.product(batch_shape), hb, @modes[opts[:mode]]) |> NxSignalAMD.unwrap!()

    %NxSignalAMD.DeviceTensor{ref: yref, ctx: x.ctx, shape: Tuple.insert_at(batch_shape, r - 1, n_out), type: {:f, 32}}
  end

  def fir(x, taps, opts) do
    opts = Keyword.validate!(opts, mode: :same)

    if not is_map_key(@modes, opts[:mode]) do
      raise ArgumentError, "expected mode to be one of [:full, :same, :valid], got: #{inspect(opts[:mode])}"
    end

    shape = Nx.shape(x)
    r = tuple_size(shape)
    length = elem(shape, r - 1)
    batch_shape = Tuple.delete_at(shape, r - 1)
    # f64 operands are transformed in c128 by the reference (convolution.ex:276-284): the f64 tier
    {type, nif, es} =
      if Nx.type(x) == {:f, 64} or Nx.type(taps) == {:f, 64}, do: {:f64, &NIF.fir_f64/6, 8}, else: {:f32, &NIF.fir/6, 4}

    xb = x |> Nx.as_type(type) |> Nx.to_binary()
    hb = taps |> Nx.as_type(type) |> Nx.to_binary()
    {:ok, y} = nif.(NxSignalAMD.context(), xb, length, Tuple.product(batch_shape), hb, @modes[opts[:mode]]) |> NxSignalAMD.unwrap!()
    n_out = div(byte_size(y), es * max(Tuple.product(batch_shape), 1))
    Nx.from_binary(y, type) |> Nx.reshape(Tuple.insert_at(batch_shape, r - 1, n_out))
  end

  defp b(true), do: 1
  defp b(_), do: 0
end
