defmodule NxSignalAMD.Convolution do
  @moduledoc """
  `NxSignal.Convolution` (lib/nx_signal/convolution.ex:38-58, :87-93, :95-218, :252-347): `convolve/3`, `correlate/3`,
  `fftconvolve/3`.  `method: :direct` (the default, as in the reference) runs time-domain sums on the device, products
  accumulated in double in the BinaryBackend's window order (exact on integer-valued data, O(output x kernel) work).
  `method: :fft`: a real stream against a real 1-D filter runs the overlap-save kernel (any length, batched over leading
  axes of `a`); every other pair of operands of equal rank (complex, 2-D, 3-D ...) runs the device-side `fft_nd` fold like
  the reference.
  """
  alias NxSignalAMD.NIF

  @modes %{full: 0, same: 1, valid: 2}

  def convolve(in1, in2, opts \\ []) do
    opts = Keyword.validate!(opts, mode: :full, method: :direct)
    mode!(opts[:mode])

    case opts[:method] do
      :fft -> fftconvolve(in1, in2, mode: opts[:mode])
      :direct -> direct_convolve(in1, in2, opts[:mode])
      other -> raise ArgumentError, "expected method to be one of [:direct, :fft], got: #{inspect(other)}"
    end
  end

  @doc "See `NxSignal.Convolution.correlate/3`: `convolve(in1, conjugate(reverse(in2)), opts)`."
  def correlate(in1, in2, opts \\ []) do
    k = Nx.reverse(in2)
    k = if match?({:c, _}, Nx.type(k)), do: Nx.conjugate(k), else: k
    convolve(in1, k, opts)
  end

  # direct_convolve (lib/nx_signal/convolution.ex:95-218): rank checks as there; scalars become one-element vectors
  defp direct_convolve(in1, in2, mode) do
    rank =
      case {Nx.rank(in1), Nx.rank(in2)} do
        {0, 0} -> 0
        {0, r} -> raise ArgumentError, message: "Incompatible ranks: {0, #{r}}"
        {r, 0} -> raise ArgumentError, message: "Incompatible ranks: {#{r}, 0}"
        {r, r} -> r
        {r1, r2} ->
          raise ArgumentError,
                "NxSignal.convolve/3 requires both inputs to have the same rank or one of them to be a scalar, got #{r1} and #{r2}"
      end

    {in1, in2} = if rank == 0, do: {Nx.reshape(in1, {1}), Nx.reshape(in2, {1})}, else: {in1, in2}
    {a, a_real} = operand(in1)
    {b, b_real} = operand(in2)

    {:ok, out, out_shape} =
      NIF.convolve_direct(NxSignalAMD.context(), a, a_real, Tuple.to_list(Nx.shape(in1)), b, b_real, Tuple.to_list(Nx.shape(in2)), mode!(mode))
      |> NxSignalAMD.unwrap!()

    type = if a_real == 1 and b_real == 1, do: :f32, else: :c64
    res = Nx.from_binary(out, type) |> Nx.reshape(List.to_tuple(out_shape))
    if rank == 0, do: Nx.reshape(res, {}), else: res
  end

  def fftconvolve(in1, in2, opts \\ []) do
    opts = Keyword.validate!(opts, [:method, mode: :full])
    mode = mode!(opts[:mode])

    complex? = match?({:c, _}, Nx.type(in1)) or match?({:c, _}, Nx.type(in2))
    ctx = NxSignalAMD.context()

    cond do
      Nx.rank(in2) == 1 and Nx.rank(in1) >= 1 and not complex? ->
        # the streaming case: overlap-save FIR, `in1` may carry leading batch axes
        {batch_shape, length} = NxSignalAMD.split_last(Nx.shape(in1))
        batch = Tuple.product(batch_shape)
        # f64 operands: both transforms in c128 (the f64 tier); a mixed pair is computed entirely in double
        {type, nif, es} =
          if Nx.type(in1) == {:f, 64} or Nx.type(in2) == {:f, 64}, do: {:f64, &NIF.fir_f64/6, 8}, else: {:f32, &NIF.fir/6, 4}

        x = in1 |> Nx.as_type(type) |> Nx.to_binary()
        h = in2 |> Nx.as_type(type) |> Nx.to_binary()
        {:ok, y} = nif.(ctx, x, length, batch, h, mode) |> NxSignalAMD.unwrap!()
        n_out = div(byte_size(y), es * max(batch, 1))
        Nx.from_binary(y, type) |> Nx.reshape(Tuple.insert_at(batch_shape, tuple_size(batch_shape), n_out))

      Nx.rank(in1) != Nx.rank(in2) ->
        raise ArgumentError, "Rank of in1 and in2 must be equal."

      true ->
        {a, a_real} = operand(in1)
        {b, b_real} = operand(in2)

        {:ok, out, out_shape} =
          NIF.fftconvolve_nd(ctx, a, a_real, Tuple.to_list(Nx.shape(in1)), b, b_real, Tuple.to_list(Nx.shape(in2)), mode)
          |> NxSignalAMD.unwrap!()

        type = if a_real == 1 and b_real == 1, do: :f32, else: :c64
        Nx.from_binary(out, type) |> Nx.reshape(List.to_tuple(out_shape))
    end
  end

  defp operand(t) do
    case Nx.type(t) do
      {:c, 64} -> {Nx.to_binary(t), 0}
      {:c, _} -> raise ArgumentError, "only c64 complex tensors are supported"
      {:f, 64} -> raise ArgumentError, "f64 tensors are not supported by the MI355X path"
      _ -> {t |> Nx.as_type(:f32) |> Nx.to_binary(), 1}
    end
  end

  defp mode!(mode) when is_map_key(@modes, mode), do: @modes[mode]

  defp mode!(mode),
    do: raise(ArgumentError, "expected mode to be one of [:full, :same, :valid], got: #{inspect(mode)}")
end
